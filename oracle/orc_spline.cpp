// TEST INFRASTRUCTURE — CPU oracle (see orc_math.hpp header). PARITY UNPINNED vs real Eigen.
// SE(3) cubic B-spline (BsplineSE3.cpp, quat_ops.h; originally OpenVINS) and the per-point
// undistortion loop of ImuProcess::UndistortPcl (IMU_Processing.hpp:452-508).
#include "orc_core.hpp"
#include "orc_spline.hpp"

namespace orc {

using M4 = std::array<double, 16>;
static M4 m4_identity() { return {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; }
static M4 m4_mul(const M4 &a, const M4 &b) {
  M4 r{};
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += a[i * 4 + k] * b[k * 4 + j];
      r[i * 4 + j] = s;
    }
  return r;
}
static M3 m4_R(const M4 &a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a[i * 4 + j];
  return r;
}
static V3 m4_t(const M4 &a) { return {a[3], a[7], a[11]}; }
static M4 m4_from(const M3 &R, V3 t) {
  M4 a = m4_identity();
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) a[i * 4 + j] = R.m[i][j];
    a[i * 4 + 3] = t[i];
  }
  return a;
}
// quat_ops.h:252-257
static M4 Inv_se3(const M4 &T) {
  M3 Rt = transpose(m4_R(T));
  V3 t = (-1.0) * (Rt * m4_t(T));
  return m4_from(Rt, t);
}
// quat_ops.h:151-188
static V3 log_so3(const M3 &R) {
  double R11 = R.m[0][0], R12 = R.m[0][1], R13 = R.m[0][2];
  double R21 = R.m[1][0], R22 = R.m[1][1], R23 = R.m[1][2];
  double R31 = R.m[2][0], R32 = R.m[2][1], R33 = R.m[2][2];
  const double tr = trace(R);
  V3 omega;
  if (tr + 1.0 < 1e-10) {
    if (std::abs(R33 + 1.0) > 1e-5)
      omega = (M_PI / std::sqrt(2.0 + 2.0 * R33)) * V3{R13, R23, 1.0 + R33};
    else if (std::abs(R22 + 1.0) > 1e-5)
      omega = (M_PI / std::sqrt(2.0 + 2.0 * R22)) * V3{R12, 1.0 + R22, R32};
    else
      omega = (M_PI / std::sqrt(2.0 + 2.0 * R11)) * V3{1.0 + R11, R21, R31};
  } else {
    double magnitude;
    const double tr_3 = tr - 3.0;
    if (tr_3 < -1e-7) {
      double theta = std::acos((tr - 1.0) / 2.0);
      magnitude = theta / (2.0 * std::sin(theta));
    } else {
      magnitude = 0.5 - tr_3 / 12.0;
    }
    omega = magnitude * V3{R32 - R23, R13 - R31, R21 - R12};
  }
  return omega;
}
// quat_ops.h:190-221
static M4 exp_se3(const double vec[6]) {
  V3 w{vec[0], vec[1], vec[2]}, u{vec[3], vec[4], vec[5]};
  double theta = std::sqrt(dot(w, w));
  M3 wskew = hat(w);
  double A, B, C;
  if (theta < 1e-7) {
    A = 1;
    B = 0.5;
    C = 1.0 / 6.0;
  } else {
    A = std::sin(theta) / theta;
    B = (1 - std::cos(theta)) / (theta * theta);
    C = (1 - A) / (theta * theta);
  }
  M3 w2 = wskew * wskew;
  M3 V = M3::I() + B * wskew + C * w2;
  M3 R = M3::I() + A * wskew + B * w2;
  return m4_from(R, V * u);
}
// quat_ops.h:224-243
static void log_se3(const M4 &mat, double out[6]) {
  V3 w = log_so3(m4_R(mat));
  V3 T = m4_t(mat);
  const double t = norm(w);
  if (t < 1e-10) {
    out[0] = w.x, out[1] = w.y, out[2] = w.z, out[3] = T.x, out[4] = T.y, out[5] = T.z;
  } else {
    M3 W = hat((1.0 / t) * w);
    double Tan = std::tan(0.5 * t);
    V3 WT = W * T;
    V3 u = T - (0.5 * t) * WT + (1 - t / (2. * Tan)) * (W * WT);
    out[0] = w.x, out[1] = w.y, out[2] = w.z, out[3] = u.x, out[4] = u.y, out[5] = u.z;
  }
}
// quat_ops.h:87-92 (JPL); BsplineSE3.cpp:41 transposes it -> Hamilton R(q) of the (x,y,z,w) state quat
static M3 quat_2_Rot(const double q[4]) {
  V3 qv{q[0], q[1], q[2]};
  M3 qx = hat(qv);
  M3 R = (2 * std::pow(q[3], 2) - 1) * M3::I() + (-2 * q[3]) * qx;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R.m[i][j] += 2 * qv[i] * qv[j];
  return R;
}

using CP = std::vector<std::pair<double, M4>>;
// BsplineSE3.cpp:121-171 on a sorted, unique-key vector (std::map semantics)
static bool find_bounding_poses(double timestamp, const CP &poses, double &t0, M4 &pose0, double &t1, M4 &pose1) {
  t0 = -1, t1 = -1;
  pose0 = m4_identity(), pose1 = m4_identity();
  bool found_older = false, found_newer = false;
  auto cmp = [](const std::pair<double, M4> &a, double v) { return a.first < v; };
  auto cmpu = [](double v, const std::pair<double, M4> &a) { return v < a.first; };
  auto lower = std::lower_bound(poses.begin(), poses.end(), timestamp, cmp);
  auto upper = std::upper_bound(poses.begin(), poses.end(), timestamp, cmpu);
  if (lower != poses.end()) {
    if (lower->first == timestamp) {
      found_older = true;
    } else if (lower != poses.begin()) {
      --lower;
      found_older = true;
    }
  }
  if (upper != poses.end()) found_newer = true;
  if (found_older) t0 = lower->first, pose0 = lower->second;
  if (found_newer) t1 = upper->first, pose1 = upper->second;
  return found_older && found_newer;
}

// BsplineSE3.cpp:26-82
void Spline::feed_trajectory(const std::vector<std::array<double, 8>> &traj_points) {
  dt = 0.01;  // :34 — both ternary arms are 0.01
  CP trajectory_points;
  for (size_t i = 0; i + 1 < traj_points.size(); i++) {  // :39 (the last sample is dropped)
    M3 R = transpose(quat_2_Rot(&traj_points[i][4]));
    V3 p{traj_points[i][1], traj_points[i][2], traj_points[i][3]};
    trajectory_points.push_back({traj_points[i][0], m4_from(R, p)});
  }
  std::stable_sort(trajectory_points.begin(), trajectory_points.end(),
                   [](const auto &a, const auto &b) { return a.first < b.first; });
  // std::map::insert keeps the FIRST value of a duplicated key
  CP uniq;
  for (auto &e : trajectory_points)
    if (uniq.empty() || uniq.back().first != e.first) uniq.push_back(e);
  trajectory_points.swap(uniq);
  double timestamp_min = INFINITY;
  for (auto &e : trajectory_points)
    if (e.first <= timestamp_min) timestamp_min = e.first;
  control_points.clear();
  double timestamp_curr = timestamp_min;
  while (true) {  // :60-77
    double t0, t1;
    M4 pose0, pose1;
    if (!find_bounding_poses(timestamp_curr, trajectory_points, t0, pose0, t1, pose1)) break;
    double lambda = (timestamp_curr - t0) / (t1 - t0);
    double lg[6];
    log_se3(m4_mul(pose1, Inv_se3(pose0)), lg);
    for (int k = 0; k < 6; k++) lg[k] *= lambda;
    M4 pose_interp = m4_mul(exp_se3(lg), pose0);
    control_points.push_back({timestamp_curr, pose_interp});
    timestamp_curr += dt;
  }
  timestamp_start = timestamp_min + 2 * dt;
}

// BsplineSE3.cpp:84-118 + :173-230
bool Spline::get_pose(double timestamp, Q &q_GtoI, V3 &p_IinG) const {
  double t0, t1, t2, t3;
  M4 pose0, pose1, pose2, pose3;
  bool success = find_bounding_poses(timestamp, control_points, t1, pose1, t2, pose2);
  if (success) {
    auto cmp = [](const std::pair<double, M4> &a, double v) { return a.first < v; };
    auto it1 = std::lower_bound(control_points.begin(), control_points.end(), t1, cmp);  // find(t1)
    auto it2 = std::lower_bound(control_points.begin(), control_points.end(), t2, cmp);  // find(t2)
    if (it1 == control_points.begin()) success = false;
    if (success) {
      auto it0 = it1 - 1;
      auto it3 = it2 + 1;
      if (it3 == control_points.end())
        success = false;
      else {
        t0 = it0->first, pose0 = it0->second;
        t3 = it3->first, pose3 = it3->second;
      }
    }
  }
  if (!success) {
    p_IinG = V3{0, 0, 0};
    return false;
  }
  double DT = (t2 - t1);
  double u = (timestamp - t1) / DT;
  double b0 = 1.0 / 6.0 * (5 + 3 * u - 3 * u * u + u * u * u);
  double b1 = 1.0 / 6.0 * (1 + 3 * u + 3 * u * u - 2 * u * u * u);
  double b2 = 1.0 / 6.0 * (u * u * u);
  double l0[6], l1[6], l2[6];
  log_se3(m4_mul(Inv_se3(pose0), pose1), l0);
  log_se3(m4_mul(Inv_se3(pose1), pose2), l1);
  log_se3(m4_mul(Inv_se3(pose2), pose3), l2);
  for (int k = 0; k < 6; k++) l0[k] *= b0, l1[k] *= b1, l2[k] *= b2;
  M4 A0 = exp_se3(l0), A1 = exp_se3(l1), A2 = exp_se3(l2);
  M4 pose_interp = m4_mul(m4_mul(m4_mul(pose0, A0), A1), A2);  // :111
  q_GtoI = fromR(m4_R(pose_interp));                            // :112-113
  p_IinG = m4_t(pose_interp);
  (void)t0;
  (void)t3;
  return true;
}

// IMU_Processing.hpp:452-508 for one LiDAR `num`.
void undistort_lidar(std::vector<Pt> &pts, double lidar_beg_time, double lidar_end_time, const Spline &spline,
                     const std::vector<double> &imu_cov_t, const std::vector<std::array<double, 36>> &imu_cov_c,
                     const Pose &extrinsic, const Pose &lt_lidar_frame, std::vector<Pose> &uncertainty) {
  int cov_pointer = (int)imu_cov_t.size() - 1;  // :453-467
  int idx = -1;
  while (true) {
    if (imu_cov_t[cov_pointer] > lidar_end_time) {
      cov_pointer = cov_pointer - 1;
    } else {
      cov_pointer = cov_pointer + 1;
      break;
    }
  }
  const Q lt_q = lt_lidar_frame.q_;
  const V3 lt_t = lt_lidar_frame.t_;
  if (pts.empty()) return;
  for (long it = (long)pts.size() - 1; it != 0; it--) {  // :475-476: end()-1 ... begin()+1
    Pt &p = pts[it];
    V3 pt_t{0, 0, 0};
    Q pt_q;
    double point_t = p.curvature / double(1000) + lidar_beg_time;  // :482
    bool spline_flag = spline.get_pose(point_t, pt_q, pt_t);
    if (imu_cov_t[cov_pointer] > point_t) {  // :484-494 (single-step `if`)
      cov_pointer = cov_pointer - 1;
      Pose pt_imu_frame, pos_calculated;
      double c[6][6];
      std::memcpy(c, imu_cov_c[cov_pointer + 1].data(), sizeof(c));
      PoseInitial(pt_imu_frame, pt_t, pt_q, c);
      compoundPoseWithCov(pt_imu_frame, pt_imu_frame.cov_, extrinsic, extrinsic.cov_, pos_calculated, pos_calculated.cov_);
      compoundInvPoseWithCov(lt_lidar_frame, lt_lidar_frame.cov_, pos_calculated, pos_calculated.cov_, pos_calculated,
                             pos_calculated.cov_);
      compoundInvPoseWithCov(extrinsic, extrinsic.cov_, pos_calculated, pos_calculated.cov_, pos_calculated,
                             pos_calculated.cov_);
      uncertainty.push_back(pos_calculated);
      idx += 1;
    }
    if (spline_flag) {  // :496-505
      V3 P_i{p.x, p.y, p.z};
      V3 T_ei = pt_t - lt_t;
      V3 P_compensate =
          conj(extrinsic.q_) * (conj(lt_q) * (pt_q * (extrinsic.q_ * P_i + extrinsic.t_) + T_ei) - extrinsic.t_);
      p.x = (float)P_compensate.x;
      p.y = (float)P_compensate.y;
      p.z = (float)P_compensate.z;
      p.intensity = (float)idx;
    }
  }
}

}  // namespace orc
