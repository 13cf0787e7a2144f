// TEST INFRASTRUCTURE - pin of the CPU oracle against the reference's own Eigen-typed code.
//
// Compiles, UNMODIFIED and where they lie under /root/reference/MA_LIO, the pieces of the hot path the oracle
// (oracle/orc_*.cpp) restates without Eigen:
//   esti_plane<float>                      include/common_lib.h:143-190      (extracted by line range, see Makefile)
//   Barfoot compounding, evalPointUncertainty  include/associate_uct.hpp:7-175   (extracted by line range)
//   esekf::update_iterated_dyn_share_modified, esekf::predict   include/IKFoM_toolkit/esekfom/esekfom.hpp (whole header,
//   with the mtk/ manifold types and src/use-ikfom.hpp: state_ikfom for lid_num = 3, get_f / df_dx / df_dw)
// common_lib.h and associate_uct.hpp cannot be included as files (they pull in ROS, PCL and tf headers), so the Makefile
// cuts the cited line ranges out of them into oracle/_ref/gen/*.inc at build time; nothing of the reference is copied
// into the repository. Needs Eigen 3 and the Boost headers (preprocessor, bind, math) - neither is installed in the
// build image of rounds 1-2, where this target is skipped and parity stays "unpinned" (DESIGN.md §1).
//
// Usage: ref_eigen <inputs.bin> <outputs.bin>   (container format: tests/golden/eigen_io.py)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>
#include <Eigen/Eigen>

using namespace std;
using namespace Eigen;

// ---- the handful of names the extracted line ranges expect from the top of common_lib.h (:19-39,57-63) ------------
#define NUM_MATCH_POINTS (5)
struct PointType {  // pcl::PointXYZINormal: the fields esti_plane / evalPointUncertainty touch
  float x, y, z, pad0, normal_x, normal_y, normal_z, pad1, intensity, curvature, pad2, pad3;
};
typedef vector<PointType, Eigen::aligned_allocator<PointType>> PointVector;
typedef Vector3d V3D;
typedef Matrix3d M3D;
typedef Eigen::Matrix<double, 6, 6> M6D;
#include "gen/common_lib_pose.inc"        // struct Pose, common_lib.h:57-63
#include "gen/so3_math_skew.inc"          // skewSymmetric / so3 helpers associate_uct.hpp relies on (so3_math.h)
#include "gen/common_lib_esti_plane.inc"  // template <typename T> bool esti_plane(...), common_lib.h:143-190
#include "gen/associate_uct.inc"          // associate_uct.hpp:7-175

#include "use-ikfom.hpp"  // -I /root/reference/MA_LIO/src -I /root/reference/MA_LIO/include

// ---- named-array container ---------------------------------------------------------------------------------------
struct Arr {
  uint32_t dtype = 1;  // 0 f32, 1 f64, 2 i32
  vector<uint32_t> dims;
  vector<char> raw;
  size_t count() const {
    size_t c = 1;
    for (auto d : dims) c *= d;
    return c;
  }
  const double *f64() const { return (const double *)raw.data(); }
  const float *f32() const { return (const float *)raw.data(); }
  const int32_t *i32() const { return (const int32_t *)raw.data(); }
};
static map<string, Arr> read_all(const char *path) {
  map<string, Arr> m;
  ifstream f(path, ios::binary);
  uint32_t nl;
  while (f.read((char *)&nl, 4)) {
    string name(nl, ' ');
    f.read(&name[0], nl);
    Arr a;
    uint32_t nd;
    f.read((char *)&a.dtype, 4), f.read((char *)&nd, 4);
    a.dims.resize(nd);
    f.read((char *)a.dims.data(), 4 * nd);
    a.raw.resize(a.count() * (a.dtype == 1 ? 8 : 4));
    f.read(a.raw.data(), a.raw.size());
    m[name] = a;
  }
  return m;
}
static void put(ofstream &f, const string &name, uint32_t dtype, vector<uint32_t> dims, const void *data) {
  uint32_t nl = name.size(), nd = dims.size();
  size_t c = 1;
  for (auto d : dims) c *= d;
  f.write((const char *)&nl, 4), f.write(name.data(), nl), f.write((const char *)&dtype, 4), f.write((const char *)&nd, 4);
  f.write((const char *)dims.data(), 4 * nd), f.write((const char *)data, c * (dtype == 1 ? 8 : 4));
}

// pose59 = q(x,y,z,w) t(3) T(4x4 row-major) cov(6x6 row-major): the layout of malio_pose_t / oracle poses
static Pose pose_from(const double *p) {
  Pose o;
  o.q_ = Quaterniond(p[3], p[0], p[1], p[2]);
  o.t_ = Vector3d(p[4], p[5], p[6]);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) o.T_(i, j) = p[7 + 4 * i + j];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) o.cov_(i, j) = p[23 + 6 * i + j];
  return o;
}
static void pose_to(const Pose &o, double *p) {
  p[0] = o.q_.x(), p[1] = o.q_.y(), p[2] = o.q_.z(), p[3] = o.q_.w();
  for (int k = 0; k < 3; k++) p[4 + k] = o.t_(k);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) p[7 + 4 * i + j] = o.T_(i, j);
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) p[23 + 6 * i + j] = o.cov_(i, j);
}
// flat state (L = 3): pos rot(x,y,z,w) offset_R[3] offset_T[3] vel bg ba grav = 40 doubles
static state_ikfom state_from(const double *s) {
  state_ikfom x;
  for (int k = 0; k < 3; k++) x.pos[k] = s[k];
  x.rot.coeffs() << s[3], s[4], s[5], s[6];
  x.offset_R_0.coeffs() << s[7], s[8], s[9], s[10];
  x.offset_R_1.coeffs() << s[11], s[12], s[13], s[14];
  x.offset_R_2.coeffs() << s[15], s[16], s[17], s[18];
  for (int k = 0; k < 3; k++) {
    x.offset_T_0[k] = s[19 + k], x.offset_T_1[k] = s[22 + k], x.offset_T_2[k] = s[25 + k];
    x.vel[k] = s[28 + k], x.bg[k] = s[31 + k], x.ba[k] = s[34 + k], x.grav.vec[k] = s[37 + k];
  }
  return x;
}
static void state_to(const state_ikfom &x, double *s) {
  for (int k = 0; k < 3; k++) s[k] = x.pos[k];
  auto q = [&](const SO3 &r, double *o) { o[0] = r.x(), o[1] = r.y(), o[2] = r.z(), o[3] = r.w(); };
  q(x.rot, s + 3), q(x.offset_R_0, s + 7), q(x.offset_R_1, s + 11), q(x.offset_R_2, s + 15);
  for (int k = 0; k < 3; k++) {
    s[19 + k] = x.offset_T_0[k], s[22 + k] = x.offset_T_1[k], s[25 + k] = x.offset_T_2[k];
    s[28 + k] = x.vel[k], s[31 + k] = x.bg[k], s[34 + k] = x.ba[k], s[37 + k] = x.grav.vec[k];
  }
}

// replayed measurement model: pass k of the update hands back the k-th recorded (valid, h_x, h, R)
static const map<string, Arr> *g_in = nullptr;
static int g_pass = 0;
static vector<int> g_converge_seen;
static void h_replay(state_ikfom &, esekfom::dyn_share_datastruct<double> &d) {
  const map<string, Arr> &in = *g_in;
  const int k = g_pass++;
  g_converge_seen.push_back(d.converge ? 1 : 0);
  const int valid = in.at("upd_valid").i32()[k], M = in.at("upd_M").i32()[k];
  if (!valid) {
    d.valid = false;
    return;
  }
  const Arr &hx = in.at("upd_hx_" + to_string(k)), &h = in.at("upd_h_" + to_string(k)), &R = in.at("upd_R_" + to_string(k));
  const int C = (int)hx.dims[1];
  d.h_x = MatrixXd::Zero(M, C), d.h.resize(M), d.R = MatrixXd::Zero(M, 1);
  for (int r = 0; r < M; r++) {
    for (int c = 0; c < C; c++) d.h_x(r, c) = hx.f64()[(size_t)r * C + c];
    d.h(r) = h.f64()[r], d.R(r, 0) = R.f64()[r];
  }
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  map<string, Arr> in = read_all(argv[1]);
  g_in = &in;
  ofstream out(argv[2], ios::binary);
  {  // ---- a3: esti_plane<float> ----
    const Arr &pl = in.at("plane_pts");  // [K][5][4] f32: x y z normal_y
    const int K = (int)pl.dims[0];
    const float th = in.at("plane_th").f32()[0];
    const double covth = in.at("cov_threshold").f64()[0];
    vector<float> pabcd((size_t)K * 4);
    vector<double> pcov(K);
    vector<int32_t> ok(K);
    for (int k = 0; k < K; k++) {
      PointVector pv(5);
      for (int j = 0; j < 5; j++) {
        const float *p = pl.f32() + ((size_t)k * 5 + j) * 4;
        memset(&pv[j], 0, sizeof(PointType));
        pv[j].x = p[0], pv[j].y = p[1], pv[j].z = p[2], pv[j].normal_y = p[3];
      }
      Matrix<float, 4, 1> r;
      double pc = 0;
      ok[k] = esti_plane(r, pv, th, pc, covth) ? 1 : 0;
      for (int j = 0; j < 4; j++) pabcd[(size_t)k * 4 + j] = r(j);
      pcov[k] = pc;
    }
    put(out, "plane_pabcd", 0, {(uint32_t)K, 4}, pabcd.data());
    put(out, "plane_cov", 1, {(uint32_t)K}, pcov.data());
    put(out, "plane_ok", 2, {(uint32_t)K}, ok.data());
  }
  {  // ---- a6: evalPointUncertainty ----
    const Arr &pt = in.at("unc_pts"), &ps = in.at("unc_poses");  // [K][3] f32, [K][59] f64
    const int K = (int)pt.dims[0];
    vector<double> cov((size_t)K * 9);
    for (int k = 0; k < K; k++) {
      PointType p;
      memset(&p, 0, sizeof(p));
      p.x = pt.f32()[3 * k], p.y = pt.f32()[3 * k + 1], p.z = pt.f32()[3 * k + 2];
      Matrix3d c;
      evalPointUncertainty(p, c, pose_from(ps.f64() + (size_t)k * 59));
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) cov[(size_t)k * 9 + 3 * i + j] = c(i, j);
    }
    put(out, "unc_cov", 1, {(uint32_t)K, 3, 3}, cov.data());
  }
  {  // ---- a15: compoundPoseWithCov / compoundInvPoseWithCov (method 2 = 4th order, as laserMapping.cpp:1028-1048) ----
    const Arr &a = in.at("comp_a"), &b = in.at("comp_b");  // [K][59]
    const int K = (int)a.dims[0];
    vector<double> o1((size_t)K * 59), o2((size_t)K * 59);
    for (int k = 0; k < K; k++) {
      Pose p1 = pose_from(a.f64() + (size_t)k * 59), p2 = pose_from(b.f64() + (size_t)k * 59), c1, c2;
      compoundPoseWithCov(p1, p1.cov_, p2, p2.cov_, c1, c1.cov_, 2);
      compoundInvPoseWithCov(p1, p1.cov_, p2, p2.cov_, c2, c2.cov_, 2);
      pose_to(c1, &o1[(size_t)k * 59]), pose_to(c2, &o2[(size_t)k * 59]);
    }
    put(out, "comp_out", 1, {(uint32_t)K, 59}, o1.data());
    put(out, "comp_inv_out", 1, {(uint32_t)K, 59}, o2.data());
  }
  {  // ---- a10/a11/a12: update_iterated_dyn_share_modified on replayed rows ----
    const int max_iter = in.at("upd_max_iter").i32()[0];
    esekfom::esekf<state_ikfom, 12, input_ikfom> kf;
    kf.init_dyn_share(get_f, df_dx, df_dw, h_replay, max_iter);
    state_ikfom x = state_from(in.at("upd_state").f64());
    esekfom::esekf<state_ikfom, 12, input_ikfom>::cov P;
    for (int i = 0; i < 35; i++)
      for (int j = 0; j < 35; j++) P(i, j) = in.at("upd_P").f64()[35 * i + j];
    kf.change_x(x), kf.change_P(P);
    double solve = 0;
    g_pass = 0;
    kf.update_iterated_dyn_share_modified(in.at("upd_Rscalar").f64()[0], solve);
    double s40[40], Pout[35 * 35];
    state_to(kf.get_x(), s40);
    for (int i = 0; i < 35; i++)
      for (int j = 0; j < 35; j++) Pout[35 * i + j] = kf.get_P()(i, j);
    int32_t passes = g_pass;
    put(out, "upd_state_out", 1, {40}, s40);
    put(out, "upd_P_out", 1, {35, 35}, Pout);
    put(out, "upd_passes", 2, {1}, &passes);
    put(out, "upd_converge_seen", 2, {(uint32_t)g_converge_seen.size()}, g_converge_seen.data());
  }
  {  // ---- f-3: esekf::predict chain ----
    const Arr &st = in.at("pred_steps");  // [K][7]: dt acc(3) gyro(3)
    const int K = (int)st.dims[0];
    esekfom::esekf<state_ikfom, 12, input_ikfom> kf;
    kf.init_dyn_share(get_f, df_dx, df_dw, h_replay, 1);
    state_ikfom x = state_from(in.at("pred_state").f64());
    esekfom::esekf<state_ikfom, 12, input_ikfom>::cov P;
    for (int i = 0; i < 35; i++)
      for (int j = 0; j < 35; j++) P(i, j) = in.at("pred_P").f64()[35 * i + j];
    kf.change_x(x), kf.change_P(P);
    Eigen::Matrix<double, 12, 12> Q;
    for (int i = 0; i < 12; i++)
      for (int j = 0; j < 12; j++) Q(i, j) = in.at("pred_Q").f64()[12 * i + j];
    vector<double> so((size_t)K * 40), Po((size_t)K * 35 * 35);
    for (int k = 0; k < K; k++) {
      const double *r = st.f64() + (size_t)k * 7;
      double dt = r[0];
      input_ikfom u;
      for (int a = 0; a < 3; a++) u.acc[a] = r[1 + a], u.gyro[a] = r[4 + a];
      kf.predict(dt, Q, u);
      state_to(kf.get_x(), &so[(size_t)k * 40]);
      for (int i = 0; i < 35; i++)
        for (int j = 0; j < 35; j++) Po[(size_t)k * 1225 + 35 * i + j] = kf.get_P()(i, j);
    }
    put(out, "pred_state_out", 1, {(uint32_t)K, 40}, so.data());
    put(out, "pred_P_out", 1, {(uint32_t)K, 35, 35}, Po.data());
  }
  return 0;
}
