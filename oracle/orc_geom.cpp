// TEST INFRASTRUCTURE — CPU oracle (see orc_math.hpp header). PARITY UNPINNED vs real Eigen.
// Plane fit, point uncertainty, SE(3) covariance compounding, k-NN providers.
#include <dlfcn.h>
#include <array>
#include "orc_core.hpp"

namespace orc {

// ---------------------------------------------------------------------------------------------
// Eigen 3.3.7 ColPivHouseholderQR<Matrix<float,5,3>> (Eigen/src/QR/ColPivHouseholderQR.h:
// computeInPlace + _solve_impl; Householder.h: makeHouseholder / applyHouseholderOnTheLeft),
// restated in float32. Called by esti_plane via A.colPivHouseholderQr().solve(b)
// (common_lib.h:174). Sums are accumulated sequentially (Eigen's internal reduction order is
// not part of its contract); compile with -ffp-contract=off so no FMA is formed, matching the
// reference's plain -O3 x86-64 build (CMakeLists.txt:8).
static void colpiv_qr_solve_5x3(float A[5][3], const float b[5], float x[3]) {
  const int rows = 5, cols = 3, size = 3;
  float hCoeffs[3];
  int transp[3];
  float normsUpdated[3], normsDirect[3];
  for (int k = 0; k < cols; k++) {
    float s = 0.f;
    for (int i = 0; i < rows; i++) s += A[i][k] * A[i][k];
    normsDirect[k] = std::sqrt(s);
    normsUpdated[k] = normsDirect[k];
  }
  const float eps = 1.1920929e-07f;  // NumTraits<float>::epsilon()
  float maxn = normsUpdated[0];
  for (int k = 1; k < cols; k++)
    if (normsUpdated[k] > maxn) maxn = normsUpdated[k];
  float th = maxn * eps;
  float threshold_helper = (th * th) / float(rows);
  float norm_downdate_threshold = std::sqrt(eps);
  int nonzero_pivots = size;
  for (int k = 0; k < size; k++) {
    int big = k;
    float bigv = normsUpdated[k];
    for (int j = k + 1; j < cols; j++)
      if (normsUpdated[j] > bigv) bigv = normsUpdated[j], big = j;
    float big_sq = bigv * bigv;
    if (nonzero_pivots == size && big_sq < threshold_helper * float(rows - k)) nonzero_pivots = k;
    transp[k] = big;
    if (k != big) {
      for (int i = 0; i < rows; i++) std::swap(A[i][k], A[i][big]);
      std::swap(normsUpdated[k], normsUpdated[big]);
      std::swap(normsDirect[k], normsDirect[big]);
    }
    // makeHouseholderInPlace on A[k:rows, k]
    float tailSq = 0.f;
    for (int i = k + 1; i < rows; i++) tailSq += A[i][k] * A[i][k];
    float c0 = A[k][k];
    float tau, beta;
    const float tol = 1.17549435e-38f;  // numeric_limits<float>::min()
    if (tailSq <= tol) {
      tau = 0.f;
      beta = c0;
      for (int i = k + 1; i < rows; i++) A[i][k] = 0.f;
    } else {
      beta = std::sqrt(c0 * c0 + tailSq);
      if (c0 >= 0.f) beta = -beta;
      float den = c0 - beta;
      for (int i = k + 1; i < rows; i++) A[i][k] = A[i][k] / den;
      tau = (beta - c0) / beta;
    }
    hCoeffs[k] = tau;
    A[k][k] = beta;
    // applyHouseholderOnTheLeft to A[k:rows, k+1:cols]
    if (rows - k == 1) {
      for (int j = k + 1; j < cols; j++) A[k][j] *= (1.f - tau);
    } else if (tau != 0.f) {
      for (int j = k + 1; j < cols; j++) {
        float tmp = 0.f;
        for (int i = k + 1; i < rows; i++) tmp += A[i][k] * A[i][j];
        tmp += A[k][j];
        A[k][j] -= tau * tmp;
        for (int i = k + 1; i < rows; i++) A[i][j] -= tau * A[i][k] * tmp;
      }
    }
    // column-norm downdate (LAPACK Working Note 176)
    for (int j = k + 1; j < cols; j++) {
      if (normsUpdated[j] != 0.f) {
        float temp = std::fabs(A[k][j]) / normsUpdated[j];
        temp = (1.f + temp) * (1.f - temp);
        temp = temp < 0.f ? 0.f : temp;
        float r = normsUpdated[j] / normsDirect[j];
        float temp2 = temp * (r * r);
        if (temp2 <= norm_downdate_threshold) {
          float s = 0.f;
          for (int i = k + 1; i < rows; i++) s += A[i][j] * A[i][j];
          normsDirect[j] = std::sqrt(s);
          normsUpdated[j] = normsDirect[j];
        } else {
          normsUpdated[j] *= std::sqrt(temp);
        }
      }
    }
  }
  // column permutation from transpositions
  int perm[3] = {0, 1, 2};
  for (int k = 0; k < size; k++) std::swap(perm[k], perm[transp[k]]);
  x[0] = x[1] = x[2] = 0.f;
  if (nonzero_pivots == 0) return;
  float c[5];
  for (int i = 0; i < rows; i++) c[i] = b[i];
  for (int k = 0; k < nonzero_pivots; k++) {  // c <- Q^T c
    float tau = hCoeffs[k];
    if (rows - k == 1) {
      c[k] *= (1.f - tau);
    } else if (tau != 0.f) {
      float tmp = 0.f;
      for (int i = k + 1; i < rows; i++) tmp += A[i][k] * c[i];
      tmp += c[k];
      c[k] -= tau * tmp;
      for (int i = k + 1; i < rows; i++) c[i] -= tau * A[i][k] * tmp;
    }
  }
  for (int i = nonzero_pivots - 1; i >= 0; i--) {  // back substitution
    float s = c[i];
    for (int j = i + 1; j < nonzero_pivots; j++) s -= A[i][j] * c[j];
    c[i] = s / A[i][i];
  }
  for (int i = 0; i < nonzero_pivots; i++) x[perm[i]] = c[i];
}

// common_lib.h:144-190
bool esti_plane(float pca_result[4], const std::vector<Pt> &point, float threshold, double &plane_cov,
                double cov_threshold) {
  float A[5][3], b[5], W[5];
  double cov_sum = 0;
  plane_cov = 0;
  for (int j = 0; j < NUM_MATCH_POINTS; j++) {  // :159-166
    A[j][0] = point[j].x;
    A[j][1] = point[j].y;
    A[j][2] = point[j].z;
    b[j] = -1.0f;
    W[j] = point[j].normal_y;
    cov_sum += std::abs(cov_threshold - W[j]);
  }
  if (W[0] > 0.00001) {  // :167-173
    for (int j = 0; j < NUM_MATCH_POINTS; j++)
      plane_cov += ((cov_threshold - W[j]) / cov_sum) * ((cov_threshold - W[j]) / cov_sum) * W[j];
  }
  float normvec[3];
  colpiv_qr_solve_5x3(A, b, normvec);  // :174
  float n = std::sqrt(normvec[0] * normvec[0] + normvec[1] * normvec[1] + normvec[2] * normvec[2]);  // :176
  pca_result[0] = normvec[0] / n;
  pca_result[1] = normvec[1] / n;
  pca_result[2] = normvec[2] / n;
  pca_result[3] = (float)(1.0 / n);  // :180 (double division, stored to float)
  for (int j = 0; j < NUM_MATCH_POINTS; j++) {  // :182-188
    if (std::fabs(pca_result[0] * point[j].x + pca_result[1] * point[j].y + pca_result[2] * point[j].z +
                  pca_result[3]) > threshold)
      return false;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// associate_uct.hpp:145-175. cov_point = (G * blkdiag(1e4*pose.cov_, 0.1*I3) * G^T)[0:3,0:3],
// G = [ pointToFS(T*p) | T*D ] (4x9).
void evalPointUncertainty(const Pt &pi, double cov_point[3][3], const Pose &pose) {
  double cov_input[9][9] = {};
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) cov_input[i][j] = pose.cov_[i][j] * 10000;
  for (int i = 0; i < 3; i++) cov_input[6 + i][6 + i] = 0.1;
  double distance_weight = 0.05;
  double pc[4] = {pi.x * distance_weight, pi.y * distance_weight, pi.z * distance_weight, 1};
  double Tp[4];
  for (int i = 0; i < 4; i++) {
    double s = 0;
    for (int k = 0; k < 4; k++) s += pose.T_[i][k] * pc[k];
    Tp[i] = s;
  }
  double G[4][9] = {};
  for (int i = 0; i < 3; i++) G[i][i] = Tp[3];  // pointToFS :148
  M3 sk = hat(V3{Tp[0], Tp[1], Tp[2]});
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) G[i][3 + j] = -sk.m[i][j];  // :149
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 3; j++) G[i][6 + j] = pose.T_[i][j];  // T * D :173
  double GS[4][9];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 9; j++) {
      double s = 0;
      for (int k = 0; k < 9; k++) s += G[i][k] * cov_input[k][j];
      GS[i][j] = s;
    }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 9; k++) s += GS[i][k] * G[j][k];
      cov_point[i][j] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// common_lib.h:129-142
void PoseInitial(Pose &pose, V3 trans, Q quat, const double cov[6][6]) {
  M3 R = toR(quat);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) pose.T_[i][j] = R.m[i][j];
    pose.T_[i][3] = trans[i];
    pose.T_[3][i] = 0;
  }
  pose.T_[3][3] = 1;
  pose.t_ = trans;
  pose.q_ = quat;
  std::memcpy(pose.cov_, cov, sizeof(double) * 36);
}

static void set_T(Pose &p) {  // associate_uct.hpp:38-42 / :93-97
  M3 R = toR(p.q_);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) p.T_[i][j] = R.m[i][j];
    p.T_[i][3] = p.t_[i];
  }
  p.T_[3][0] = p.T_[3][1] = p.T_[3][2] = 0;
  p.T_[3][3] = 1;
}

// adjointMatrix(T.inverse()) (associate_uct.hpp:8-15,44,99). T is rigid, so the 4x4 inverse Eigen
// computes equals [R^T, -R^T t] to rounding.
static Mat adjoint_of_inverse(const double T[4][4]) {
  M3 Rt;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Rt.m[i][j] = T[j][i];
  V3 t{T[0][3], T[1][3], T[2][3]};
  V3 ti = (-1.0) * (Rt * t);
  M3 tr = hat(ti) * Rt;
  Mat Ad(6, 6);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Ad(i, j) = Rt.m[i][j];
      Ad(i, 3 + j) = tr.m[i][j];
      Ad(3 + i, 3 + j) = Rt.m[i][j];
    }
  return Ad;
}
static M3 blk(const Mat &A, int r, int c) {
  M3 m;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) m.m[i][j] = A(r + i, c + j);
  return m;
}
static void setblk(Mat &A, int r, int c, const M3 &m) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A(r + i, c + j) = m.m[i][j];
}
static M3 covop1(const M3 &B) { return (-trace(B)) * M3::I() + B; }                                  // :17-21
static M3 covop2(const M3 &B, const M3 &C) { return covop1(B) * covop1(C) + covop1(C * B); }          // :23-27

// 4th-order terms shared by both compound functions (associate_uct.hpp:53-81 == :106-134)
static void fourth_order(const Mat &cov_1_prime, const Mat &cov_2, double cov_cp[6][6]) {
  M3 c1rr = blk(cov_1_prime, 0, 0), c1rp = blk(cov_1_prime, 0, 3), c1pp = blk(cov_1_prime, 3, 3);
  M3 c2rr = blk(cov_2, 0, 0), c2rp = blk(cov_2, 0, 3), c2pp = blk(cov_2, 3, 3);
  Mat A1(6, 6), A2(6, 6), B(6, 6);
  setblk(A1, 0, 0, covop1(c1pp));
  setblk(A1, 0, 3, covop1(c1rp + transpose(c1rp)));
  setblk(A1, 3, 3, covop1(c1pp));
  setblk(A2, 0, 0, covop1(c2pp));
  setblk(A2, 0, 3, covop1(c2rp + transpose(c2rp)));
  setblk(A2, 3, 3, covop1(c2pp));
  M3 Brr = covop2(c1pp, c2rr) + covop2(transpose(c1rp), c2rp) + covop2(c1rp, transpose(c2rp)) + covop2(c1rr, c2pp);
  M3 Brp = covop2(c1pp, transpose(c2rp)) + covop2(transpose(c1rp), c2pp);
  M3 Bpp = covop2(c1pp, c2pp);
  setblk(B, 0, 0, Brr);
  setblk(B, 0, 3, Brp);
  setblk(B, 3, 0, transpose(Brp));
  setblk(B, 3, 3, Bpp);
  Mat S = A1 * cov_2 + cov_2 * transpose(A1) + A2 * cov_1_prime + cov_1_prime * transpose(A2);
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) cov_cp[i][j] = cov_1_prime(i, j) + cov_2(i, j) + S(i, j) / 12 + B(i, j) / 4;
}
static Mat to_mat6(const double c[6][6]) {
  Mat m(6, 6);
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) m(i, j) = c[i][j];
  return m;
}

// associate_uct.hpp:29-83. Aliasing (pose_cp == pose_2, cov_cp == cov_2) is what the callers do
// (IMU_Processing.hpp:490-491, laserMapping.cpp:1044); inputs are snapshotted in reference order.
void compoundInvPoseWithCov(const Pose &pose_1, const double cov_1[6][6], const Pose &pose_2,
                            const double cov_2[6][6], Pose &pose_cp, double cov_cp[6][6]) {
  Mat c1 = to_mat6(cov_1), c2 = to_mat6(cov_2);
  Q q = conj(pose_1.q_) * pose_2.q_;                // :35
  V3 t = conj(pose_1.q_) * (pose_2.t_ - pose_1.t_);  // :36
  pose_cp.q_ = q;
  pose_cp.t_ = t;
  set_T(pose_cp);                                // :38-42
  Mat AdT = adjoint_of_inverse(pose_cp.T_);      // :44
  Mat c1p = AdT * c1 * transpose(AdT);           // :45
  fourth_order(c1p, c2, cov_cp);                 // :53-81
}

// associate_uct.hpp:85-142. NOTE the reference evaluates adjointMatrix(pose_2.T_.inverse()) AFTER
// writing pose_cp.T_ (:95-99); when pose_cp aliases pose_2 (laserMapping.cpp:1043,
// IMU_Processing.hpp — not :489 where the output is a fresh Pose) the COMPOSED transform is used.
void compoundPoseWithCov(const Pose &pose_1, const double cov_1[6][6], const Pose &pose_2, const double cov_2[6][6],
                         Pose &pose_cp, double cov_cp[6][6]) {
  Mat c1 = to_mat6(cov_1), c2 = to_mat6(cov_2);
  Q q = pose_1.q_ * pose_2.q_;                 // :90
  V3 t = pose_1.q_ * pose_2.t_ + pose_1.t_;    // :91
  pose_cp.q_ = q;
  pose_cp.t_ = t;
  set_T(pose_cp);                              // :93-97 (may overwrite pose_2.T_ when aliased)
  Mat AdT2 = adjoint_of_inverse(pose_2.T_);    // :99
  Mat c1p = AdT2 * c1 * transpose(AdT2);       // :100
  fourth_order(c1p, c2, cov_cp);               // :106-134
  if (cov_cp != pose_cp.cov_) std::memcpy(pose_cp.cov_, cov_cp, sizeof(double) * 36);  // :135
}

// ---------------------------------------------------------------------------------------------
// Independent exact k-NN (static k-d tree, float32 arithmetic as ikd_Tree.cpp:1693-1720).
struct KdKnn : Knn {
  struct Node {
    int lo, hi, left = -1, right = -1;
    float bmin[3], bmax[3];
  };
  std::vector<Pt> pts;
  std::vector<Node> nodes;
  static float box_dist(const Node &nd, const Pt &p) {  // calc_box_dist, ikd_Tree.cpp:1702-1720
    float d = 0.f;
    const float q[3] = {p.x, p.y, p.z};
    for (int a = 0; a < 3; a++) {
      if (q[a] < nd.bmin[a]) d += (q[a] - nd.bmin[a]) * (q[a] - nd.bmin[a]);
      if (q[a] > nd.bmax[a]) d += (q[a] - nd.bmax[a]) * (q[a] - nd.bmax[a]);
    }
    return d;
  }
  int build_rec(int lo, int hi) {
    Node nd;
    nd.lo = lo, nd.hi = hi;
    for (int a = 0; a < 3; a++) nd.bmin[a] = INFINITY, nd.bmax[a] = -INFINITY;
    for (int i = lo; i < hi; i++) {
      const float q[3] = {pts[i].x, pts[i].y, pts[i].z};
      for (int a = 0; a < 3; a++) nd.bmin[a] = std::min(nd.bmin[a], q[a]), nd.bmax[a] = std::max(nd.bmax[a], q[a]);
    }
    int id = (int)nodes.size();
    nodes.push_back(nd);
    if (hi - lo > 16) {
      int ax = 0;
      float ext = nd.bmax[0] - nd.bmin[0];
      for (int a = 1; a < 3; a++)
        if (nd.bmax[a] - nd.bmin[a] > ext) ext = nd.bmax[a] - nd.bmin[a], ax = a;
      int mid = (lo + hi) / 2;
      std::nth_element(pts.begin() + lo, pts.begin() + mid, pts.begin() + hi,
                       [ax](const Pt &a, const Pt &b) { return (&a.x)[ax] < (&b.x)[ax]; });
      int l = build_rec(lo, mid);
      int r = build_rec(mid, hi);
      nodes[id].left = l, nodes[id].right = r;
    }
    return id;
  }
  void build(const std::vector<Pt> &p) override {
    pts = p;
    nodes.clear();
    if (!pts.empty()) build_rec(0, (int)pts.size());
  }
  int size() override { return (int)pts.size(); }
  struct Cand {
    float d;
    int i;
  };
  void rec(int id, const Pt &q, int k, std::vector<Cand> &best) {
    const Node &nd = nodes[id];
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi; i++) {
        float dx = q.x - pts[i].x, dy = q.y - pts[i].y, dz = q.z - pts[i].z;
        float d = dx * dx + dy * dy + dz * dz;  // calc_dist ikd_Tree.cpp:1697
        if ((int)best.size() < k || d < best.back().d) {
          Cand c{d, i};
          auto it = std::upper_bound(best.begin(), best.end(), c, [](const Cand &a, const Cand &b) { return a.d < b.d; });
          best.insert(it, c);
          if ((int)best.size() > k) best.pop_back();
        }
      }
      return;
    }
    float dl = box_dist(nodes[nd.left], q), dr = box_dist(nodes[nd.right], q);
    int first = nd.left, second = nd.right;
    float d2nd = dr;
    if (dr < dl) first = nd.right, second = nd.left, d2nd = dl;
    rec(first, q, k, best);
    if ((int)best.size() < k || d2nd < best.back().d) rec(second, q, k, best);
  }
  void search(const Pt &q, int k, std::vector<Pt> &near, std::vector<float> &d2) override {
    std::vector<Cand> best;
    best.reserve(k + 1);
    if (!nodes.empty()) rec(0, q, k, best);
    near.clear();
    d2.clear();
    for (auto &c : best) near.push_back(pts[c.i]), d2.push_back(c.d);
  }
};
Knn *make_kd_knn() { return new KdKnn(); }

// The reference's own ikd-Tree through oracle/_ref/libikd_ref.so (built by oracle/ref_ikdtree/Makefile).
struct RefKnn : Knn {
  void *so = nullptr, *tree = nullptr;
  void *(*f_create)(float) = nullptr;
  void (*f_destroy)(void *) = nullptr;
  void (*f_build)(void *, const float *, int) = nullptr;
  int (*f_size)(void *) = nullptr;
  int (*f_knn)(void *, const float *, int, int, float *, float *, int *, int) = nullptr;
  float ds = 0.5f;
  ~RefKnn() override {
    if (tree) f_destroy(tree);
  }
  void build(const std::vector<Pt> &p) override {
    if (tree) f_destroy(tree);
    tree = f_create(ds);
    f_build(tree, (const float *)p.data(), (int)p.size());
  }
  int size() override { return tree ? f_size(tree) : 0; }
  void search(const Pt &q, int k, std::vector<Pt> &near, std::vector<float> &d2) override {
    std::vector<Pt> out(k);
    std::vector<float> d(k);
    int cnt = 0;
    f_knn(tree, (const float *)&q, 1, k, (float *)out.data(), d.data(), &cnt, 1);
    near.assign(out.begin(), out.begin() + cnt);
    d2.assign(d.begin(), d.begin() + cnt);
  }
};
Knn *make_ref_knn(const char *so_path, float downsample) {
  void *so = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
  if (!so) return nullptr;
  RefKnn *r = new RefKnn();
  r->so = so;
  r->ds = downsample;
  r->f_create = (void *(*)(float))dlsym(so, "refikd_create");
  r->f_destroy = (void (*)(void *))dlsym(so, "refikd_destroy");
  r->f_build = (void (*)(void *, const float *, int))dlsym(so, "refikd_build");
  r->f_size = (int (*)(void *))dlsym(so, "refikd_size");
  r->f_knn = (int (*)(void *, const float *, int, int, float *, float *, int *, int))dlsym(so, "refikd_knn");
  if (!r->f_create || !r->f_destroy || !r->f_build || !r->f_size || !r->f_knn) {
    delete r;
    return nullptr;
  }
  return r;
}

}  // namespace orc
