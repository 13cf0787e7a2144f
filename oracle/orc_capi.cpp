// TEST INFRASTRUCTURE — CPU oracle (see orc_math.hpp header). Flat C API for ctypes
// (tests/, __graft_entry__.smoke(), bench.py cpu_baseline only).
//
// Flat layouts:
//   point  : 12 floats, pcl::PointXYZINormal memory layout (x y z _ nx ny nz _ intensity curvature _ _)
//   pose   : 59 doubles = q(x,y,z,w) t(3) T(16 row-major) cov(36 row-major)       (common_lib.h:57-63)
//   state  : pos(3) rot(x,y,z,w) offset_R[L](4 each) offset_T[L](3 each) vel bg ba grav   = 19+7L doubles
//   params : 17 doubles = lid_num max_iteration extrinsic_est_en plane_th cov_threshold range_min
//            range_max point_cov_max point_cov_min plane_cov_max plane_cov_min localize_cov_max
//            localize_cov_min localize_thresh_max localize_thresh_min filter_size_map limit (0 = 0.001)
#include <omp.h>
#include "orc_core.hpp"
#include "orc_spline.hpp"

using namespace orc;

static Pose pose_from(const double *p) {
  Pose r;
  r.q_ = Q{p[0], p[1], p[2], p[3]};
  r.t_ = V3{p[4], p[5], p[6]};
  std::memcpy(r.T_, p + 7, sizeof(double) * 16);
  std::memcpy(r.cov_, p + 23, sizeof(double) * 36);
  return r;
}
static void pose_to(const Pose &r, double *p) {
  p[0] = r.q_.x, p[1] = r.q_.y, p[2] = r.q_.z, p[3] = r.q_.w;
  p[4] = r.t_.x, p[5] = r.t_.y, p[6] = r.t_.z;
  std::memcpy(p + 7, r.T_, sizeof(double) * 16);
  std::memcpy(p + 23, r.cov_, sizeof(double) * 36);
}
static State state_from(const double *s, int L) {
  State x;
  x.L = L;
  const double *p = s;
  x.pos = V3{p[0], p[1], p[2]}, p += 3;
  x.rot = Q{p[0], p[1], p[2], p[3]}, p += 4;
  for (int l = 0; l < L; l++) x.offset_R[l] = Q{p[0], p[1], p[2], p[3]}, p += 4;
  for (int l = 0; l < L; l++) x.offset_T[l] = V3{p[0], p[1], p[2]}, p += 3;
  x.vel = V3{p[0], p[1], p[2]}, p += 3;
  x.bg = V3{p[0], p[1], p[2]}, p += 3;
  x.ba = V3{p[0], p[1], p[2]}, p += 3;
  x.grav = V3{p[0], p[1], p[2]};
  return x;
}
static void state_to(const State &x, double *s) {
  double *p = s;
  auto v3 = [&](V3 v) { p[0] = v.x, p[1] = v.y, p[2] = v.z, p += 3; };
  auto q4 = [&](Q q) { p[0] = q.x, p[1] = q.y, p[2] = q.z, p[3] = q.w, p += 4; };
  v3(x.pos), q4(x.rot);
  for (int l = 0; l < x.L; l++) q4(x.offset_R[l]);
  for (int l = 0; l < x.L; l++) v3(x.offset_T[l]);
  v3(x.vel), v3(x.bg), v3(x.ba), v3(x.grav);
}

struct Handle {
  Scene sc;
  Knn *knn = nullptr;
  bool is_ref = false;
};

extern "C" {

void *orc_create(const double *prm, int threads, const char *ref_so) {
  Handle *h = new Handle();
  Params &p = h->sc.prm;
  p.lid_num = (int)prm[0], p.max_iteration = (int)prm[1], p.extrinsic_est_en = (int)prm[2];
  p.plane_th = (float)prm[3], p.cov_threshold = prm[4], p.range_min = prm[5], p.range_max = prm[6];
  p.point_cov_max = prm[7], p.point_cov_min = prm[8], p.plane_cov_max = prm[9], p.plane_cov_min = prm[10];
  p.localize_cov_max = prm[11], p.localize_cov_min = prm[12], p.localize_thresh_max = prm[13];
  p.localize_thresh_min = prm[14], p.filter_size_map = prm[15];
  p.limit = prm[16] > 0 ? prm[16] : 0.001;
  h->sc.threads = threads < 1 ? 1 : threads;
  if (ref_so && ref_so[0]) {
    h->knn = make_ref_knn(ref_so, (float)p.filter_size_map);
    h->is_ref = h->knn != nullptr;
  }
  if (!h->knn) h->knn = make_kd_knn();
  h->sc.knn = h->knn;
  return h;
}
void orc_destroy(void *hh) {
  Handle *h = (Handle *)hh;
  delete h->knn;
  delete h;
}
// Replay mode for the Eigen pin (oracle/ref_eigen): npass recorded measurement results, rows of pass k =
// hx[off_k .. off_k + M_k) with off_k = sum of the earlier M; valid[k] == 0 -> ekfom_data.valid = false.
void orc_set_replay(void *hh, int npass, const int *valid, const int *M, const double *hx, const double *hv, const double *Rv) {
  Handle *h = (Handle *)hh;
  const int C = 6 * (1 + h->sc.prm.lid_num);
  h->sc.replay.clear(), h->sc.replay_pos = 0;
  size_t off = 0;
  for (int k = 0; k < npass; k++) {
    DynShare d;
    d.valid = valid[k] != 0;
    d.h_x = Mat(M[k], C);
    for (int r = 0; r < M[k]; r++)
      for (int c = 0; c < C; c++) d.h_x(r, c) = hx[(off + r) * C + c];
    d.h.assign(hv + off, hv + off + M[k]), d.R.assign(Rv + off, Rv + off + M[k]);
    off += M[k];
    h->sc.replay.push_back(d);
  }
}
void orc_set_pass_hook(void *hh, void (*fn)(int, void *), void *user) {
  Handle *h = (Handle *)hh;
  h->sc.pass_hook = fn, h->sc.pass_hook_user = user;
}
int orc_is_ref(void *hh) { return ((Handle *)hh)->is_ref ? 1 : 0; }
void orc_set_threads(void *hh, int t) { ((Handle *)hh)->sc.threads = t < 1 ? 1 : t; }

int orc_map_build(void *hh, const float *pts12, int n) {
  Handle *h = (Handle *)hh;
  std::vector<Pt> v(n);
  if (n) std::memcpy((void *)v.data(), pts12, sizeof(Pt) * (size_t)n);
  h->knn->build(v);
  return h->knn->size();
}

// batched Nearest_Search through the scene's provider, with the reference's OMP pattern
int orc_knn(void *hh, const float *q12, int nq, int k, float *out12, float *out_d2, int *out_cnt) {
  Handle *h = (Handle *)hh;
  omp_set_num_threads(h->sc.threads);
#pragma omp parallel for
  for (int i = 0; i < nq; i++) {
    Pt q;
    std::memcpy((void *)&q, q12 + (size_t)i * 12, sizeof(Pt));
    std::vector<Pt> near;
    std::vector<float> d2;
    h->knn->search(q, k, near, d2);
    out_cnt[i] = (int)near.size();
    for (int j = 0; j < k; j++) {
      if (j < (int)near.size()) {
        std::memcpy(out12 + ((size_t)i * k + j) * 12, (void *)&near[j], sizeof(Pt));
        out_d2[(size_t)i * k + j] = d2[j];
      } else {
        std::memset(out12 + ((size_t)i * k + j) * 12, 0, sizeof(Pt));
        out_d2[(size_t)i * k + j] = INFINITY;
      }
    }
  }
  return 0;
}

int orc_scan_set(void *hh, const float *pts12, int n, const int *table_len, const double *tables, const double *tc) {
  Handle *h = (Handle *)hh;
  std::vector<Pt> v(n);
  if (n) std::memcpy((void *)v.data(), pts12, sizeof(Pt) * (size_t)n);
  h->sc.set_scan(v);
  int L = h->sc.prm.lid_num;
  h->sc.pose_unc.assign(L, {});
  const double *p = tables;
  for (int l = 0; l < L; l++)
    for (int k = 0; k < table_len[l]; k++, p += 59) h->sc.pose_unc[l].push_back(pose_from(p));
  h->sc.temporal_comp.clear();
  for (int l = 0; l + 1 < L; l++) h->sc.temporal_comp.push_back(pose_from(tc + 59 * l));
  return 0;
}

// map_incremental selection (laserMapping.cpp:398-442). world_normal_y [N] or null: feats_down_world[i].normal_y as the
// caller's cloud holds it. out_add12 / out_non12 need capacity N*12 floats; counts[0] = |PointToAdd|, counts[1] =
// |PointNoNeedDownsample|.
int orc_map_incremental(void *hh, const double *state, int flg_EKF_inited, const float *world_normal_y, float *out_add12,
                        float *out_non12, int *counts) {
  Handle *h = (Handle *)hh;
  State s = state_from(state, h->sc.prm.lid_num);
  const size_t n = h->sc.feats_down_body.size();
  for (size_t i = 0; i < n; i++) h->sc.feats_down_world[i].normal_y = world_normal_y ? world_normal_y[i] : 0.f;
  std::vector<Pt> a, b;
  h->sc.map_incremental(s, flg_EKF_inited != 0, a, b);
  if (!a.empty()) std::memcpy(out_add12, (void *)a.data(), sizeof(Pt) * a.size());
  if (!b.empty()) std::memcpy(out_non12, (void *)b.data(), sizeof(Pt) * b.size());
  counts[0] = (int)a.size(), counts[1] = (int)b.size();
  return 0;
}

// One h_share_model pass. hx/h/R need capacity N*C / N / N. Returns M (0 when !valid).
int orc_h_share_model(void *hh, const double *state, int converge, int *valid, double *hx, double *hv, double *Rv,
                      double *weight) {
  Handle *h = (Handle *)hh;
  State s = state_from(state, h->sc.prm.lid_num);
  DynShare d;
  d.valid = true;
  d.converge = converge != 0;
  h->sc.h_share_model(s, d);
  *valid = d.valid ? 1 : 0;
  if (!d.valid) return 0;
  int M = d.h_x.r;
  if (hx) std::memcpy(hx, d.h_x.a.data(), sizeof(double) * d.h_x.a.size());
  if (hv) std::memcpy(hv, d.h.data(), sizeof(double) * M);
  if (Rv) std::memcpy(Rv, d.R.data(), sizeof(double) * M);
  if (weight) *weight = h->sc.last_weight;
  return M;
}

// Multi-GPU test support: local extrema of the last pass; override with all-reduced values (null = off).
void orc_last_minmax(void *hh, double *out4) { std::memcpy(out4, ((Handle *)hh)->sc.last_minmax, sizeof(double) * 4); }
void orc_set_override(void *hh, const double *mm4, int skip_loc_weight) {
  Scene &sc = ((Handle *)hh)->sc;
  sc.use_override = mm4 != nullptr;
  if (mm4) std::memcpy(sc.override_minmax, mm4, sizeof(double) * 4);
  sc.skip_loc_weight = skip_loc_weight != 0;
}

// Side effects later code relies on. Any pointer may be null.
void orc_scan_get(void *hh, float *normal_y, float *nearest12, int *nearest_cnt, unsigned char *selected,
                  float *res_last, float *world_xyz, float *normvec4) {
  Handle *h = (Handle *)hh;
  Scene &sc = h->sc;
  size_t n = sc.feats_down_body.size();
  for (size_t i = 0; i < n; i++) {
    if (normal_y) normal_y[i] = sc.feats_down_body[i].normal_y;
    if (nearest_cnt) nearest_cnt[i] = (int)sc.Nearest_Points[i].size();
    if (nearest12)
      for (int j = 0; j < 5; j++) {
        if (j < (int)sc.Nearest_Points[i].size())
          std::memcpy(nearest12 + (i * 5 + j) * 12, (void *)&sc.Nearest_Points[i][j], sizeof(Pt));
        else
          std::memset(nearest12 + (i * 5 + j) * 12, 0, sizeof(Pt));
      }
    if (selected) selected[i] = sc.point_selected_surf[i] ? 1 : 0;
    if (res_last) res_last[i] = sc.res_last[i];
    if (world_xyz) {
      world_xyz[i * 3] = sc.feats_down_world[i].x, world_xyz[i * 3 + 1] = sc.feats_down_world[i].y;
      world_xyz[i * 3 + 2] = sc.feats_down_world[i].z;
    }
    if (normvec4) {
      normvec4[i * 4] = sc.normvec[i].x, normvec4[i * 4 + 1] = sc.normvec[i].y, normvec4[i * 4 + 2] = sc.normvec[i].z;
      normvec4[i * 4 + 3] = sc.normvec[i].intensity;
    }
  }
}

// esekfom.hpp:495-721. state/P in-out. stats = {passes, searches, last_M}. trace (optional) receives the
// state after every pass' boxplus, (19+7L) doubles each, capacity max_iteration+1 entries.
int orc_update_iterated(void *hh, double *state, double *P, double R, int *stats, double *trace, double *solve_time) {
  Handle *h = (Handle *)hh;
  int L = h->sc.prm.lid_num;
  State x = state_from(state, L);
  int n = x.dof();
  Mat Pm(n, n);
  std::memcpy(Pm.a.data(), P, sizeof(double) * n * n);
  UpdateStats st;
  std::vector<State> tr;
  update_iterated(h->sc, x, Pm, R, st, trace ? &tr : nullptr);
  state_to(x, state);
  std::memcpy(P, Pm.a.data(), sizeof(double) * n * n);
  if (stats) stats[0] = st.passes, stats[1] = st.searches, stats[2] = st.last_M;
  if (trace)
    for (size_t i = 0; i < tr.size(); i++) state_to(tr[i], trace + i * (19 + 7 * L));
  if (solve_time) *solve_time = st.solve_time;
  return (int)tr.size();
}

// esekfom.hpp:388-492 + use-ikfom.hpp:67-112. state/P in-out, Q 12x12 row-major.
void orc_predict(int L, double *state, double *P, double dt, const double *Q, const double *acc, const double *gyro) {
  State x = state_from(state, L);
  int n = x.dof();
  Mat Pm(n, n), Qm(12, 12);
  std::memcpy(Pm.a.data(), P, sizeof(double) * n * n);
  std::memcpy(Qm.a.data(), Q, sizeof(double) * 144);
  predict(x, Pm, dt, Qm, V3{acc[0], acc[1], acc[2]}, V3{gyro[0], gyro[1], gyro[2]});
  state_to(x, state);
  std::memcpy(P, Pm.a.data(), sizeof(double) * n * n);
}

// ---- unit-level entry points -----------------------------------------------------------------
int orc_esti_plane(const float *near12, float threshold, double cov_threshold, float *pabcd, double *plane_cov) {
  std::vector<Pt> v(5);
  std::memcpy((void *)v.data(), near12, sizeof(Pt) * 5);
  return esti_plane(pabcd, v, threshold, *plane_cov, cov_threshold) ? 1 : 0;
}
void orc_eval_point_uncertainty(const float *p12, const double *pose59, double *cov9) {
  Pt p;
  std::memcpy((void *)&p, p12, sizeof(Pt));
  Pose ps = pose_from(pose59);
  double c[3][3];
  evalPointUncertainty(p, c, ps);
  std::memcpy(cov9, c, sizeof(c));
}
// inverse != 0: compoundInvPoseWithCov, else compoundPoseWithCov. alias != 0: output aliases pose_2
// as at laserMapping.cpp:1043-1044 / IMU_Processing.hpp:490-491.
void orc_compound(const double *pose1, const double *pose2, int inverse, int alias, double *out59) {
  Pose p1 = pose_from(pose1), p2 = pose_from(pose2);
  if (alias) {
    if (inverse)
      compoundInvPoseWithCov(p1, p1.cov_, p2, p2.cov_, p2, p2.cov_);
    else
      compoundPoseWithCov(p1, p1.cov_, p2, p2.cov_, p2, p2.cov_);
    pose_to(p2, out59);
  } else {
    Pose o;
    if (inverse)
      compoundInvPoseWithCov(p1, p1.cov_, p2, p2.cov_, o, o.cov_);
    else
      compoundPoseWithCov(p1, p1.cov_, p2, p2.cov_, o, o.cov_);
    pose_to(o, out59);
  }
}
void orc_boxplus(double *state, int L, const double *dx) {
  State x = state_from(state, L);
  std::vector<double> d(dx, dx + x.dof());
  boxplus(x, d);
  state_to(x, state);
}
void orc_boxminus(const double *state, const double *other, int L, double *res) {
  State x = state_from(state, L), o = state_from(other, L);
  std::vector<double> r;
  boxminus(x, o, r);
  std::memcpy(res, r.data(), sizeof(double) * r.size());
}

// ---- spline / undistortion ---------------------------------------------------------------------
void *orc_spline_create(const double *traj8, int n) {
  Spline *s = new Spline();
  std::vector<std::array<double, 8>> t(n);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 8; k++) t[i][k] = traj8[i * 8 + k];
  s->feed_trajectory(t);
  return s;
}
void orc_spline_destroy(void *s) { delete (Spline *)s; }
int orc_spline_num_control(void *s) { return (int)((Spline *)s)->control_points.size(); }
void orc_spline_control(void *s, double *times, double *poses16) {
  Spline *sp = (Spline *)s;
  for (size_t i = 0; i < sp->control_points.size(); i++) {
    times[i] = sp->control_points[i].first;
    std::memcpy(poses16 + i * 16, sp->control_points[i].second.data(), sizeof(double) * 16);
  }
}
int orc_spline_get_pose(void *s, double t, double *q4, double *p3) {
  Q q;
  V3 p;
  bool ok = ((Spline *)s)->get_pose(t, q, p);
  q4[0] = q.x, q4[1] = q.y, q4[2] = q.z, q4[3] = q.w;
  p3[0] = p.x, p3[1] = p.y, p3[2] = p.z;
  return ok ? 1 : 0;
}
// IMU_Processing.hpp:452-508 for one LiDAR. pts12 in/out. Returns the number of uncertainty entries
// written to unc59 (capacity unc_cap).
int orc_undistort(void *s, float *pts12, int n, double lidar_beg_time, double lidar_end_time, const double *imu_cov_t,
                  const double *imu_cov36, int n_imu, const double *extrinsic59, const double *lt_frame59,
                  double *unc59, int unc_cap) {
  std::vector<Pt> v(n);
  if (n) std::memcpy((void *)v.data(), pts12, sizeof(Pt) * (size_t)n);
  std::vector<double> t(imu_cov_t, imu_cov_t + n_imu);
  std::vector<std::array<double, 36>> c(n_imu);
  for (int i = 0; i < n_imu; i++) std::memcpy(c[i].data(), imu_cov36 + (size_t)i * 36, sizeof(double) * 36);
  Pose ext = pose_from(extrinsic59), lt = pose_from(lt_frame59);
  std::vector<Pose> unc;
  undistort_lidar(v, lidar_beg_time, lidar_end_time, *(Spline *)s, t, c, ext, lt, unc);
  if (n) std::memcpy(pts12, (void *)v.data(), sizeof(Pt) * (size_t)n);
  for (size_t i = 0; i < unc.size() && (int)i < unc_cap; i++) pose_to(unc[i], unc59 + i * 59);
  return (int)unc.size();
}
}
