// Host side of the continuous-time trajectory (SE(3) cubic B-spline, originally OpenVINS) the undistortion
// kernel evaluates per raw point. Mirrors ov_core::BsplineSE3 of the reference
// (/root/reference/MA_LIO/src/BsplineSE3.cpp:26-118,121-230; include/quat_ops.h:87-92,151-257):
//   malio_spline_feed      == BsplineSE3::feed_trajectory  (uniform 10 ms control poses by SE(3) lerp)
//   malio_spline_get_pose  == BsplineSE3::get_pose         (needed on the host for the scan-end poses,
//                                                            IMU_Processing.hpp:430,470,483)
// plus spline_interval_logs(): log_se3(T_k^-1 T_{k+1}) per knot interval, which get_pose recomputes for every
// point (three log_se3 per call) although it only depends on the interval - the device kernel reads them.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "../csrc/malio_internal.hpp"

namespace malio {

struct SE3 {
  double R[9];  // row-major
  double t[3];
};
static SE3 se3_identity() { return {{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}}; }
static SE3 se3_mul(const SE3 &a, const SE3 &b) {
  SE3 r;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) r.R[i * 3 + j] = a.R[i * 3] * b.R[j] + a.R[i * 3 + 1] * b.R[3 + j] + a.R[i * 3 + 2] * b.R[6 + j];
    r.t[i] = a.R[i * 3] * b.t[0] + a.R[i * 3 + 1] * b.t[1] + a.R[i * 3 + 2] * b.t[2] + a.t[i];
  }
  return r;
}
static SE3 se3_inv(const SE3 &a) {  // quat_ops.h:252-257
  SE3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.R[i * 3 + j] = a.R[j * 3 + i];
  for (int i = 0; i < 3; i++) r.t[i] = -(r.R[i * 3] * a.t[0] + r.R[i * 3 + 1] * a.t[1] + r.R[i * 3 + 2] * a.t[2]);
  return r;
}
static SE3 from16(const double *T) {
  SE3 r;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) r.R[i * 3 + j] = T[i * 4 + j];
    r.t[i] = T[i * 4 + 3];
  }
  return r;
}
static void to16(const SE3 &a, double *T) {
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T[i * 4 + j] = a.R[i * 3 + j];
    T[i * 4 + 3] = a.t[i];
  }
  T[12] = T[13] = T[14] = 0, T[15] = 1;
}
// quat_ops.h:151-188
static void log_so3(const double *R, double w[3]) {
  const double tr = R[0] + R[4] + R[8];
  if (tr + 1.0 < 1e-10) {
    double s;
    if (std::fabs(R[8] + 1.0) > 1e-5) {
      s = M_PI / std::sqrt(2.0 + 2.0 * R[8]);
      w[0] = s * R[2], w[1] = s * R[5], w[2] = s * (1.0 + R[8]);
    } else if (std::fabs(R[4] + 1.0) > 1e-5) {
      s = M_PI / std::sqrt(2.0 + 2.0 * R[4]);
      w[0] = s * R[1], w[1] = s * (1.0 + R[4]), w[2] = s * R[7];
    } else {
      s = M_PI / std::sqrt(2.0 + 2.0 * R[0]);
      w[0] = s * (1.0 + R[0]), w[1] = s * R[3], w[2] = s * R[6];
    }
    return;
  }
  double magnitude;
  const double tr_3 = tr - 3.0;
  if (tr_3 < -1e-7) {
    double theta = std::acos((tr - 1.0) / 2.0);
    magnitude = theta / (2.0 * std::sin(theta));
  } else {
    magnitude = 0.5 - tr_3 / 12.0;
  }
  w[0] = magnitude * (R[7] - R[5]), w[1] = magnitude * (R[2] - R[6]), w[2] = magnitude * (R[3] - R[1]);
}
// quat_ops.h:224-243
static void log_se3(const SE3 &m, double out[6]) {
  double w[3];
  log_so3(m.R, w);
  const double t = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  out[0] = w[0], out[1] = w[1], out[2] = w[2];
  if (t < 1e-10) {
    out[3] = m.t[0], out[4] = m.t[1], out[5] = m.t[2];
    return;
  }
  const double a[3] = {w[0] / t, w[1] / t, w[2] / t};
  auto crs = [](const double *x, const double *y, double *z) {
    z[0] = x[1] * y[2] - x[2] * y[1], z[1] = x[2] * y[0] - x[0] * y[2], z[2] = x[0] * y[1] - x[1] * y[0];
  };
  double WT[3], WWT[3];
  crs(a, m.t, WT);
  crs(a, WT, WWT);
  const double Tan = std::tan(0.5 * t);
  for (int k = 0; k < 3; k++) out[3 + k] = m.t[k] - (0.5 * t) * WT[k] + (1 - t / (2. * Tan)) * WWT[k];
}
// quat_ops.h:190-221
static SE3 exp_se3(const double v[6]) {
  const double *w = v, *u = v + 3;
  const double theta = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double A, B, C;
  if (theta < 1e-7) {
    A = 1, B = 0.5, C = 1.0 / 6.0;
  } else {
    A = std::sin(theta) / theta;
    B = (1 - std::cos(theta)) / (theta * theta);
    C = (1 - A) / (theta * theta);
  }
  const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double K2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) K2[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
  SE3 r;
  double V[9];
  for (int i = 0; i < 9; i++) {
    double I = (i % 4 == 0) ? 1.0 : 0.0;
    r.R[i] = I + A * K[i] + B * K2[i];
    V[i] = I + B * K[i] + C * K2[i];
  }
  for (int i = 0; i < 3; i++) r.t[i] = V[i * 3] * u[0] + V[i * 3 + 1] * u[1] + V[i * 3 + 2] * u[2];
  return r;
}

// index of the knot interval that get_pose uses for `ts` (BsplineSE3.cpp:121-230 on std::map semantics):
// i1 = last knot <= ts, needs knots i1-1 and i1+2. Returns -1 when the reference's get_pose fails.
int spline_interval(const double *times, int n, double ts) {
  int i1 = (int)(std::upper_bound(times, times + n, ts) - times) - 1;  // number of knots <= ts, minus 1
  if (i1 < 1 || i1 + 2 >= n) return -1;
  return i1;
}

void spline_interval_logs(const double *poses16, int n, double *logs6 /*[(n-1)*6]*/) {
  for (int k = 0; k + 1 < n; k++) log_se3(se3_mul(se3_inv(from16(poses16 + 16 * k)), from16(poses16 + 16 * (k + 1))), logs6 + 6 * k);
}

}  // namespace malio

using namespace malio;

extern "C" {

// BsplineSE3::feed_trajectory (BsplineSE3.cpp:26-82). traj8[n][8] = t, p(3), q(x,y,z,w). Writes the control poses.
int malio_spline_feed(const double *traj8, int n, double *out_times, double *out_poses16, int cap, int *out_n) {
  if (!traj8 || n < 2 || !out_times || !out_poses16 || !out_n) return MALIO_ERR_BAD_ARG;
  std::vector<std::pair<double, SE3>> tp;
  for (int i = 0; i + 1 < n; i++) {  // :39 drops the last sample
    const double *r = traj8 + 8 * i;
    const double x = r[4], y = r[5], z = r[6], w = r[7];
    SE3 T;  // quat_2_Rot(q)^T (:41) == Hamilton rotation matrix of (x,y,z,w)
    T.R[0] = 2 * w * w - 1 + 2 * x * x, T.R[1] = 2 * x * y - 2 * w * z, T.R[2] = 2 * x * z + 2 * w * y;
    T.R[3] = 2 * x * y + 2 * w * z, T.R[4] = 2 * w * w - 1 + 2 * y * y, T.R[5] = 2 * y * z - 2 * w * x;
    T.R[6] = 2 * x * z - 2 * w * y, T.R[7] = 2 * y * z + 2 * w * x, T.R[8] = 2 * w * w - 1 + 2 * z * z;
    T.t[0] = r[1], T.t[1] = r[2], T.t[2] = r[3];
    tp.push_back({r[0], T});
  }
  std::stable_sort(tp.begin(), tp.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
  std::vector<std::pair<double, SE3>> u;  // std::map::insert keeps the first of equal keys
  for (auto &e : tp)
    if (u.empty() || u.back().first != e.first) u.push_back(e);
  const double dt = 0.01;  // :34
  int k = 0;
  double ts = u.front().first;
  while (true) {
    // find_bounding_poses (:121-171)
    auto lo = std::lower_bound(u.begin(), u.end(), ts, [](const auto &a, double v) { return a.first < v; });
    auto up = std::upper_bound(u.begin(), u.end(), ts, [](double v, const auto &a) { return v < a.first; });
    bool older = false;
    if (lo != u.end()) {
      if (lo->first == ts)
        older = true;
      else if (lo != u.begin()) {
        --lo;
        older = true;
      }
    }
    if (!older || up == u.end()) break;
    double lambda = (ts - lo->first) / (up->first - lo->first);
    double lg[6];
    log_se3(se3_mul(up->second, se3_inv(lo->second)), lg);
    for (double &v : lg) v *= lambda;
    if (k >= cap) return MALIO_ERR_ALLOC;
    out_times[k] = ts;
    to16(se3_mul(exp_se3(lg), lo->second), out_poses16 + 16 * k);
    k++;
    ts += dt;
  }
  *out_n = k;
  return MALIO_OK;
}

// BsplineSE3::get_pose (BsplineSE3.cpp:84-118). Returns 1 on success (q as x,y,z,w), 0 when the spline cannot
// bound `timestamp` (p set to zero like the reference).
int malio_spline_get_pose(const double *times, const double *poses16, int n, double timestamp, double q[4],
                          double p[3]) {
  if (!times || !poses16 || !q || !p) return MALIO_ERR_BAD_ARG;
  int i1 = spline_interval(times, n, timestamp);
  if (i1 < 0) {
    p[0] = p[1] = p[2] = 0;
    return 0;
  }
  const double DT = times[i1 + 1] - times[i1];
  const double u = (timestamp - times[i1]) / DT;
  const double b[3] = {1.0 / 6.0 * (5 + 3 * u - 3 * u * u + u * u * u), 1.0 / 6.0 * (1 + 3 * u + 3 * u * u - 2 * u * u * u),
                       1.0 / 6.0 * (u * u * u)};
  SE3 T = from16(poses16 + 16 * (i1 - 1));
  for (int k = 0; k < 3; k++) {
    double lg[6];
    log_se3(se3_mul(se3_inv(from16(poses16 + 16 * (i1 - 1 + k))), from16(poses16 + 16 * (i1 + k))), lg);
    for (double &v : lg) v *= b[k];
    T = se3_mul(T, exp_se3(lg));
  }
  // Eigen quaternion-from-matrix
  const double *R = T.R;
  double tr = R[0] + R[4] + R[8], qq[4];
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0);
    qq[3] = 0.5 * s;
    s = 0.5 / s;
    qq[0] = (R[7] - R[5]) * s, qq[1] = (R[2] - R[6]) * s, qq[2] = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    qq[i] = 0.5 * s;
    s = 0.5 / s;
    qq[3] = (R[k * 3 + j] - R[j * 3 + k]) * s;
    qq[j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
    qq[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
  }
  for (int c = 0; c < 4; c++) q[c] = qq[c];
  for (int c = 0; c < 3; c++) p[c] = T.t[c];
  return 1;
}
}
