// TEST INFRASTRUCTURE — CPU oracle (see orc_math.hpp header). PARITY UNPINNED vs real Eigen.
// h_share_model (laserMapping.cpp:552-760) and update_iterated_dyn_share_modified
// (esekfom.hpp:495-721) restated without Eigen/PCL/ROS.
#include <omp.h>
#include <array>
#include "orc_core.hpp"

namespace orc {

void Scene::set_scan(const std::vector<Pt> &body) {
  feats_down_body = body;
  size_t n = body.size();
  feats_down_world.assign(n, Pt());  // laserMapping.cpp:1013
  normvec.assign(n, Pt());           // :1012
  laserCloudOri.assign(n, Pt());
  corr_normvect.assign(n, Pt());
  Nearest_Points.assign(n, std::vector<Pt>());  // :1025 (resize keeps old content in the reference;
                                                //  every entry is rewritten by the first pass)
  point_selected_surf.assign(n, 0);
  res_last.assign(n, 0.f);
  cov_plane.assign(n, 0.0);
}

// 3x3 symmetric eigenvalues (Jacobi) — used for the localization weight: the singular values of
// the M x 3 matrix svd_mat (laserMapping.cpp:745-747) are the square roots of the eigenvalues of
// svd_mat^T svd_mat, so sigma_3/sigma_1 is identical.
static void sym3_eig(double A[3][3], double ev[3]) {
  double a[3][3];
  std::memcpy(a, A, sizeof(a));
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (a[p][q] == 0) continue;
        double theta = (a[q][q] - a[p][p]) / (2 * a[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
        double c = 1 / std::sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < 3; k++) {
          double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {
          double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
      }
  }
  ev[0] = a[0][0], ev[1] = a[1][1], ev[2] = a[2][2];
  std::sort(ev, ev + 3);
}

// laserMapping.cpp:552-760
void Scene::h_share_model(const State &s, DynShare &ekfom_data) {
  if (!replay.empty()) {
    const DynShare &r = replay[replay_pos++ % replay.size()];
    ekfom_data.valid = r.valid, ekfom_data.h_x = r.h_x, ekfom_data.h = r.h, ekfom_data.R = r.R;
    effct_feat_num = r.h_x.r;
    return;
  }
  const int feats_down_size = (int)feats_down_body.size();
  const int lid_num = prm.lid_num;
  // extrinsic_update() (:291-308, :557): the void* state pointers alias the filter's live state,
  // so these are the ITERATED extrinsics s.offset_R/T (SURVEY.md §7 "pointer aliasing").
  std::vector<Q> extrinsic_quat(lid_num);
  std::vector<V3> extrinsic_trans(lid_num);
  for (int num = 0; num < lid_num; num++) extrinsic_quat[num] = s.offset_R[num], extrinsic_trans[num] = s.offset_T[num];

  cov_plane.resize(feats_down_size);  // :556
  omp_set_num_threads(threads);       // :560
#pragma omp parallel for
  for (int i = 0; i < feats_down_size; i++) {  // :563-612
    Pt &point_body = feats_down_body[i];
    Pt &point_world = feats_down_world[i];
    V3 p_body{point_body.x, point_body.y, point_body.z};
    int lid_idx = (int)point_body.intensity;  // :570
    if (lid_idx != 0)                         // :571-572
      p_body = conj(extrinsic_quat[0]) *
               ((temporal_comp[lid_idx - 1].q_ * (extrinsic_quat[lid_idx] * p_body + extrinsic_trans[lid_idx]) +
                 temporal_comp[lid_idx - 1].t_) -
                extrinsic_trans[0]);
    V3 p_global = s.rot * (extrinsic_quat[0] * p_body + extrinsic_trans[0]) + s.pos;  // :574
    point_world.x = (float)p_global.x;
    point_world.y = (float)p_global.y;
    point_world.z = (float)p_global.z;
    point_world.intensity = point_body.intensity;

    std::vector<float> pointSearchSqDis(NUM_MATCH_POINTS);
    auto &points_near = Nearest_Points[i];
    if (ekfom_data.converge) {  // :583-588
      knn->search(point_world, NUM_MATCH_POINTS, points_near, pointSearchSqDis);
      point_selected_surf[i] = (int)points_near.size() < NUM_MATCH_POINTS           ? false
                               : pointSearchSqDis[NUM_MATCH_POINTS - 1] > 5 ? false
                                                                            : true;
    }
    if (!point_selected_surf[i]) continue;  // :590

    float pabcd[4];
    double unit_cov;
    point_selected_surf[i] = false;
    if (esti_plane(pabcd, points_near, prm.plane_th, unit_cov, prm.cov_threshold)) {  // :596
      float pd2 = pabcd[0] * point_world.x + pabcd[1] * point_world.y + pabcd[2] * point_world.z + pabcd[3];  // :598
      float sc = (float)(1 - 0.9 * std::fabs(pd2) / std::sqrt(norm(p_body)));                                 // :599
      if (sc > 0.1) {  // :601
        point_selected_surf[i] = true;
        normvec[i].x = pabcd[0];
        normvec[i].y = pabcd[1];
        normvec[i].z = pabcd[2];
        normvec[i].intensity = pd2;
        cov_plane[i] = unit_cov;
        res_last[i] = std::abs(pd2);
      }
    }
  }

  effct_feat_num = 0;  // :614-632
  double max_unit_cov = 0;
  double min_unit_cov = 1000;
  for (int i = 0; i < feats_down_size; i++) {
    if (point_selected_surf[i]) {
      laserCloudOri[effct_feat_num] = feats_down_body[i];
      corr_normvect[effct_feat_num] = normvec[i];
      cov_plane[effct_feat_num] = cov_plane[i];
      if (cov_plane[effct_feat_num] > max_unit_cov) max_unit_cov = cov_plane[effct_feat_num];
      if (cov_plane[effct_feat_num] < min_unit_cov) min_unit_cov = cov_plane[effct_feat_num];
      effct_feat_num++;
    }
  }
  last_minmax[0] = max_unit_cov, last_minmax[1] = min_unit_cov;
  last_minmax[2] = 0, last_minmax[3] = 9999;
  if (use_override) max_unit_cov = override_minmax[0], min_unit_cov = override_minmax[1];
  if (effct_feat_num < 1) {  // :635-639
    ekfom_data.valid = false;
    return;
  }

  const int C = (1 + lid_num) * 6;
  ekfom_data.h_x = Mat(effct_feat_num, C);  // :642-644
  ekfom_data.h.assign(effct_feat_num, 0.0);
  ekfom_data.R.assign(effct_feat_num, 0.0);
  double max_cov = 0;
  double min_cov = 9999;

  for (int i = 0; i < effct_feat_num; i++) {  // :649-708
    if (cov_plane[i] == 0)
      cov_plane[i] = 1;
    else if (max_unit_cov == min_unit_cov)
      cov_plane[i] = (prm.plane_cov_max + prm.plane_cov_min) / 2;
    else
      cov_plane[i] = 1 / ((prm.plane_cov_max - prm.plane_cov_min) * (cov_plane[i] - min_unit_cov) /
                              (max_unit_cov - min_unit_cov) +
                          prm.plane_cov_min);

    double cov[3][3];
    Pt &laser_p = laserCloudOri[i];
    V3 point_this_be{laser_p.x, laser_p.y, laser_p.z};
    int lid_idx = (int)laser_p.intensity;
    if (lid_idx != 0)  // :662-663
      point_this_be =
          conj(extrinsic_quat[0]) *
          ((temporal_comp[lid_idx - 1].q_ * (extrinsic_quat[lid_idx] * point_this_be + extrinsic_trans[lid_idx]) +
            temporal_comp[lid_idx - 1].t_) -
           extrinsic_trans[0]);
    M3 point_be_crossmat = hat(point_this_be);
    V3 point_this = extrinsic_quat[0] * point_this_be + extrinsic_trans[0];  // :667
    M3 point_crossmat = hat(point_this);
    const Pt &norm_p = corr_normvect[i];
    V3 norm_vec{norm_p.x, norm_p.y, norm_p.z};
    V3 Cv = conj(s.rot) * norm_vec;  // :676
    V3 A = point_crossmat * Cv;      // :677
    V3 B;
    ekfom_data.h_x(i, 0) = norm_p.x, ekfom_data.h_x(i, 1) = norm_p.y, ekfom_data.h_x(i, 2) = norm_p.z;  // :679
    ekfom_data.h_x(i, 3) = A.x, ekfom_data.h_x(i, 4) = A.y, ekfom_data.h_x(i, 5) = A.z;
    if (prm.extrinsic_est_en) {  // :681-704
      if (lid_idx == 0) {
        B = point_be_crossmat * (conj(extrinsic_quat[0]) * Cv);  // :684 (Matrix3d * Quaternion -> R product; same map)
      } else {
        V3 point_ori{laser_p.x, laser_p.y, laser_p.z};
        point_be_crossmat = hat(point_ori);
        Cv = conj(temporal_comp[lid_idx - 1].q_) * Cv;               // :689
        B = point_be_crossmat * (conj(extrinsic_quat[lid_idx]) * Cv);  // :690
      }
      for (int k = 0; k < 3; k++) ekfom_data.h_x(i, 6 + 3 * lid_idx + k) = B[k];              // :692
      for (int k = 0; k < 3; k++) ekfom_data.h_x(i, 6 + 3 * (lid_num + lid_idx) + k) = Cv[k];  // :693
      int uncertain = int(laser_p.normal_x);                                                   // :694
      // :695 compares int with size_t (unsigned): a negative index also takes this branch
      if ((size_t)uncertain >= pose_unc[lid_idx].size()) uncertain = (int)pose_unc[lid_idx].size() - 2;
      evalPointUncertainty(laser_p, cov, pose_unc[lid_idx][uncertain]);
      ekfom_data.R[i] = cov[0][0] + cov[1][1] + cov[2][2];
      laser_p.normal_y = (float)(cov[0][0] + cov[1][1] + cov[2][2]);
      if (max_cov < ekfom_data.R[i]) max_cov = ekfom_data.R[i];
      if (min_cov > ekfom_data.R[i]) min_cov = ekfom_data.R[i];
    }
    ekfom_data.h[i] = (-1) * norm_p.intensity;  // :707
  }

  last_minmax[2] = max_cov, last_minmax[3] = min_cov;
  if (use_override) max_cov = override_minmax[2], min_cov = override_minmax[3];
  for (int i = 0; i < effct_feat_num; i++) {  // :711-722 (FIC)
    for (int j = 0; j < C; j++) ekfom_data.h_x(i, j) = ekfom_data.h_x(i, j) * cov_plane[i];
    ekfom_data.h[i] = ekfom_data.h[i] * cov_plane[i];
    if (ekfom_data.R[i] < min_cov + (max_cov - min_cov) * prm.range_min)
      ekfom_data.R[i] = prm.point_cov_min;
    else if (ekfom_data.R[i] > min_cov + (max_cov - min_cov) * prm.range_max)
      ekfom_data.R[i] = prm.point_cov_max;
    else
      ekfom_data.R[i] = (prm.point_cov_max - prm.point_cov_min) *
                            (ekfom_data.R[i] - (min_cov + (max_cov - min_cov) * prm.range_min)) /
                            ((prm.range_max - prm.range_min) * (max_cov - min_cov)) +
                        prm.point_cov_min;
  }

  int k = 0;  // :725-743
  for (int i = 0; i < feats_down_size; i++) {
    if (point_selected_surf[i]) {
      feats_down_body[i].normal_y = laserCloudOri[k].normal_y;
      k++;
    } else {
      double cov[3][3];
      int which_lidar = (int)feats_down_body[i].intensity;  // float which_lidar used as index (:736)
      int imu_idx = int(feats_down_body[i].normal_x);
      // :738 compares int with size_t: (size_t)imu_idx >= size()-1  (unsigned arithmetic)
      if ((size_t)imu_idx >= pose_unc[which_lidar].size() - 1) imu_idx = (int)pose_unc[which_lidar].size() - 2;
      evalPointUncertainty(feats_down_body[i], cov, pose_unc[which_lidar][imu_idx]);
      feats_down_body[i].normal_y = (float)(cov[0][0] + cov[1][1] + cov[2][2]);
    }
  }

  // :745-759 localization weight = sigma_3 / sigma_1 of h_x[:, 0:3]
  double NtN[3][3] = {};
  for (int i = 0; i < effct_feat_num; i++)
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) NtN[a][b] += ekfom_data.h_x(i, a) * ekfom_data.h_x(i, b);
  double ev[3];
  sym3_eig(NtN, ev);
  double weight = std::sqrt(std::max(ev[0], 0.0)) / std::sqrt(ev[2]);
  if (weight > prm.localize_thresh_max)
    weight = prm.localize_cov_max;
  else if (weight < prm.localize_thresh_min)
    weight = prm.localize_cov_min;
  else
    weight = (prm.localize_cov_max - prm.localize_cov_min) * (weight - prm.localize_thresh_min) /
                 (prm.localize_thresh_max - prm.localize_thresh_min) +
             prm.localize_cov_min;
  last_weight = weight;
  if (skip_loc_weight) return;
  for (auto &v : ekfom_data.h_x.a) v *= weight;  // :758
  for (auto &v : ekfom_data.h) v *= weight;      // :759
}

// ---------------------------------------------------------------------------------------------
// S2<double, 98090, 10000, 1> (use-ikfom.hpp:8; S2.hpp)
static const double S2_LEN = 98090.0 / 10000.0;
struct M32 {
  double m[3][2];
};
static M32 S2_Bx(V3 vec) {  // S2.hpp:189-242, S2_typ == 1 branch (:225-241)
  M32 r;
  const double length = S2_LEN;
  if (vec[0] + length > mtk_tol()) {
    r.m[0][0] = -vec[1], r.m[0][1] = -vec[2];
    r.m[1][0] = length - vec[1] * vec[1] / (length + vec[0]), r.m[1][1] = -vec[2] * vec[1] / (length + vec[0]);
    r.m[2][0] = -vec[2] * vec[1] / (length + vec[0]), r.m[2][1] = length - vec[2] * vec[2] / (length + vec[0]);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 2; j++) r.m[i][j] /= length;
  } else {
    for (int i = 0; i < 3; i++) r.m[i][0] = r.m[i][1] = 0;
    r.m[1][1] = -1;
    r.m[2][0] = 1;
  }
  return r;
}
static void S2_boxplus(V3 &vec, double d0, double d1) {  // S2.hpp:136-142
  M32 Bx = S2_Bx(vec);
  V3 Bu{Bx.m[0][0] * d0 + Bx.m[0][1] * d1, Bx.m[1][0] * d0 + Bx.m[1][1] * d1, Bx.m[2][0] * d0 + Bx.m[2][1] * d1};
  Q res = so3_exp(Bu, 1.0);  // MTK::exp(res.vec, Bu, scale/2) with scale = 1
  vec = toR(res) * vec;
}
static void S2_boxminus(V3 vec, V3 other, double res[2]) {  // S2.hpp:144-167
  double v_sin = norm(hat(vec) * other);
  double v_cos = dot(vec, other);
  double theta = std::atan2(v_sin, v_cos);
  if (v_sin < mtk_tol()) {
    if (std::fabs(theta) > mtk_tol()) {
      res[0] = 3.1415926;
      res[1] = 0;
    } else {
      res[0] = 0;
      res[1] = 0;
    }
  } else {
    M32 Bx = S2_Bx(other);
    V3 hv = hat(other) * vec;
    for (int j = 0; j < 2; j++) res[j] = theta / v_sin * (Bx.m[0][j] * hv[0] + Bx.m[1][j] * hv[1] + Bx.m[2][j] * hv[2]);
  }
}
static void S2_Nx_yy(V3 vec, double Nx[2][3]) {  // S2.hpp:269-274
  M32 Bx = S2_Bx(vec);
  M3 h = hat(vec);
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += Bx.m[k][i] * h.m[k][j];
      Nx[i][j] = 1 / S2_LEN / S2_LEN * s;
    }
}
static void S2_Mx(V3 vec, double d0, double d1, double Mx[3][2]) {  // S2.hpp:276-290
  M32 Bx = S2_Bx(vec);
  M3 h = hat(vec);
  M3 left;
  if (std::sqrt(d0 * d0 + d1 * d1) < mtk_tol()) {
    left = (-1.0) * h;
  } else {
    V3 Bu{Bx.m[0][0] * d0 + Bx.m[0][1] * d1, Bx.m[1][0] * d0 + Bx.m[1][1] * d1, Bx.m[2][0] * d0 + Bx.m[2][1] * d1};
    Q e = mtk_exp_scale(Bu, 0.0);  // scalar(1/2) == 0: integer division quirk, S2.hpp:287
    left = (-1.0) * (toR(e) * h * transpose(A_matrix(Bu)));
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 2; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += left.m[i][k] * Bx.m[k][j];
      Mx[i][j] = s;
    }
}

void boxplus(State &x, const std::vector<double> &dx) {
  const int L = x.L;
  auto v3 = [&](int o) { return V3{dx[o], dx[o + 1], dx[o + 2]}; };
  x.pos = x.pos + v3(0);                        // vect.hpp boxplus
  x.rot = x.rot * so3_exp(v3(3));               // SOn.hpp:241-244
  for (int l = 0; l < L; l++) x.offset_R[l] = x.offset_R[l] * so3_exp(v3(6 + 3 * l));
  for (int l = 0; l < L; l++) x.offset_T[l] = x.offset_T[l] + v3(6 + 3 * L + 3 * l);
  x.vel = x.vel + v3(6 + 6 * L);
  x.bg = x.bg + v3(9 + 6 * L);
  x.ba = x.ba + v3(12 + 6 * L);
  S2_boxplus(x.grav, dx[15 + 6 * L], dx[16 + 6 * L]);
}
void boxminus(const State &x, const State &o, std::vector<double> &res) {
  const int L = x.L;
  res.assign(x.dof(), 0.0);
  auto put = [&](int off, V3 v) { res[off] = v.x, res[off + 1] = v.y, res[off + 2] = v.z; };
  put(0, x.pos - o.pos);
  put(3, so3_log(conj(o.rot) * x.rot));  // SOn.hpp:245-247
  for (int l = 0; l < L; l++) put(6 + 3 * l, so3_log(conj(o.offset_R[l]) * x.offset_R[l]));
  for (int l = 0; l < L; l++) put(6 + 3 * L + 3 * l, x.offset_T[l] - o.offset_T[l]);
  put(6 + 6 * L, x.vel - o.vel);
  put(9 + 6 * L, x.bg - o.bg);
  put(12 + 6 * L, x.ba - o.ba);
  double r2[2];
  S2_boxminus(x.grav, o.grav, r2);
  res[15 + 6 * L] = r2[0], res[16 + 6 * L] = r2[1];
}

// Left-/right-multiply the rows/cols [idx, idx+d) of an n x n matrix by a d x d block.
static void rows_mul(Mat &P, const Mat &src, int idx, int d, const double *B /*d x d row-major*/, int ncols) {
  for (int c = 0; c < ncols; c++) {
    double tmp[3];
    for (int i = 0; i < d; i++) {
      double s = 0;
      for (int k = 0; k < d; k++) s += B[i * d + k] * src(idx + k, c);
      tmp[i] = s;
    }
    for (int i = 0; i < d; i++) P(idx + i, c) = tmp[i];
  }
}
static void cols_mulT(Mat &P, int idx, int d, const double *B, int nrows) {  // P[:, idx:idx+d] *= B^T
  for (int r = 0; r < nrows; r++) {
    double tmp[3];
    for (int j = 0; j < d; j++) {
      double s = 0;
      for (int k = 0; k < d; k++) s += P(r, idx + k) * B[j * d + k];
      tmp[j] = s;
    }
    for (int j = 0; j < d; j++) P(r, idx + j) = tmp[j];
  }
}

// esekfom.hpp:495-721
void update_iterated(Scene &sc, State &x_, Mat &P_, double R, UpdateStats &st, std::vector<State> *trace_states) {
  const int n = x_.dof();
  const int L = x_.L;
  const int C = 6 * (L + 1);
  const int maximum_iter = sc.prm.max_iteration;
  DynShare dyn_share;
  dyn_share.valid = true;
  dyn_share.converge = true;
  int t = 0;
  State x_propagated = x_;
  Mat P_propagated = P_;
  std::vector<int> so3_idx;  // SO3_state: rot, offset_R_l (build_manifold.hpp:113)
  so3_idx.push_back(3);
  for (int l = 0; l < L; l++) so3_idx.push_back(6 + 3 * l);
  const int s2_idx = 15 + 6 * L;

  Mat K_h(n, 1), K_x(n, n);
  std::vector<double> dx_new(n, 0.0);
  for (int i = -1; i < maximum_iter; i++) {  // :509
    dyn_share.valid = true;
    if (dyn_share.converge) st.searches++;
    if (sc.pass_hook) sc.pass_hook(st.passes, sc.pass_hook_user);
    sc.h_share_model(x_, dyn_share);  // :512
    st.passes++;
    std::vector<double> R_dyn = dyn_share.R;
    if (!dyn_share.valid) continue;  // :514-517
    const Mat &h_x_ = dyn_share.h_x;
    double solve_start = omp_get_wtime();
    const int dof_Measurement = h_x_.r;
    st.last_M = dof_Measurement;
    std::vector<double> dx;
    boxminus(x_, x_propagated, dx);  // :526
    dx_new = dx;
    P_ = P_propagated;

    for (int idx : so3_idx) {  // :534-549
      V3 seg{dx[idx], dx[idx + 1], dx[idx + 2]};
      M3 At = transpose(A_matrix(seg));
      double B[9];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) B[a * 3 + b] = At.m[a][b];
      double tmp[3];
      for (int a = 0; a < 3; a++) tmp[a] = B[a * 3] * dx_new[idx] + B[a * 3 + 1] * dx_new[idx + 1] + B[a * 3 + 2] * dx_new[idx + 2];
      for (int a = 0; a < 3; a++) dx_new[idx + a] = tmp[a];
      rows_mul(P_, P_, idx, 3, B, n);
      cols_mulT(P_, idx, 3, B, n);
    }
    {  // :551-572
      double Nx[2][3], Mx[3][2], B[4];
      S2_Nx_yy(x_.grav, Nx);
      S2_Mx(x_propagated.grav, dx[s2_idx], dx[s2_idx + 1], Mx);
      for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) B[a * 2 + b] = Nx[a][0] * Mx[0][b] + Nx[a][1] * Mx[1][b] + Nx[a][2] * Mx[2][b];
      double t0 = B[0] * dx_new[s2_idx] + B[1] * dx_new[s2_idx + 1];
      double t1 = B[2] * dx_new[s2_idx] + B[3] * dx_new[s2_idx + 1];
      dx_new[s2_idx] = t0, dx_new[s2_idx + 1] = t1;
      rows_mul(P_, P_, s2_idx, 2, B, n);
      cols_mulT(P_, s2_idx, 2, B, n);
    }

    if (n > dof_Measurement) {  // :574-582
      Mat h_x_cur(dof_Measurement, n);
      for (int r = 0; r < dof_Measurement; r++)
        for (int c = 0; c < C; c++) h_x_cur(r, c) = h_x_(r, c);
      Mat HPHt = h_x_cur * P_ * transpose(h_x_cur);
      for (auto &v : HPHt.a) v /= R;
      Mat S = HPHt + Mat::I(dof_Measurement);
      Mat K_ = P_ * transpose(h_x_cur) * inverse(S);
      for (auto &v : K_.a) v /= R;
      Mat hv(dof_Measurement, 1);
      for (int r = 0; r < dof_Measurement; r++) hv(r, 0) = dyn_share.h[r];
      K_h = K_ * hv;
      K_x = K_ * h_x_cur;
    } else {  // :621-637
      Mat P_temp = inverse(P_);
      Mat HT = transpose(h_x_);  // C x M
      for (int m = 0; m < dof_Measurement; m++) {
        if (R_dyn[m] < 0.0001) R_dyn[m] = 0.001;
        for (int c = 0; c < C; c++) HT(c, m) = HT(c, m) / R_dyn[m];
      }
      Mat HTH = HT * h_x_;
      for (int a = 0; a < C; a++)
        for (int b = 0; b < C; b++) P_temp(a, b) += HTH(a, b);
      Mat P_inv = inverse(P_temp);
      Mat hv(dof_Measurement, 1);
      for (int r = 0; r < dof_Measurement; r++) hv(r, 0) = dyn_share.h[r];
      Mat Pl(n, C);
      for (int a = 0; a < n; a++)
        for (int b = 0; b < C; b++) Pl(a, b) = P_inv(a, b);
      K_h = (Pl * HT) * hv;
      K_x = Mat(n, n);
      Mat KxL = Pl * HTH;
      for (int a = 0; a < n; a++)
        for (int b = 0; b < C; b++) K_x(a, b) = KxL(a, b);
    }

    std::vector<double> dx_(n);  // :642
    for (int a = 0; a < n; a++) {
      double sacc = K_h(a, 0);
      for (int b = 0; b < n; b++) sacc += (K_x(a, b) - (a == b ? 1.0 : 0.0)) * dx_new[b];
      dx_[a] = sacc;
    }
    boxplus(x_, dx_);  // :646
    if (trace_states) trace_states->push_back(x_);

    dyn_share.converge = true;  // :649-657 (limit[i] = 0.001 unless the caller set another one, esekfom.hpp:160-163)
    for (int a = 0; a < n; a++)
      if (std::fabs(dx_[a]) > sc.prm.limit) {
        dyn_share.converge = false;
        break;
      }
    if (dyn_share.converge) t++;
    if (!t && i == maximum_iter - 2) dyn_share.converge = true;  // :660-663

    if (t > 1 || i == maximum_iter - 1) {  // :665-718
      Mat L_ = P_;
      for (int idx : so3_idx) {
        V3 seg{dx_[idx], dx_[idx + 1], dx_[idx + 2]};
        M3 At = transpose(A_matrix(seg));
        double B[9];
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) B[a * 3 + b] = At.m[a][b];
        rows_mul(L_, P_, idx, 3, B, n);   // :676-678  L_.block(idx,i) = B * P_.block(idx,i)
        rows_mul(K_x, K_x, idx, 3, B, C);  // :679-681
        cols_mulT(L_, idx, 3, B, n);       // :682-685
        cols_mulT(P_, idx, 3, B, n);
      }
      {
        double Nx[2][3], Mx[3][2], B[4];
        S2_Nx_yy(x_.grav, Nx);
        S2_Mx(x_propagated.grav, dx_[s2_idx], dx_[s2_idx + 1], Mx);
        for (int a = 0; a < 2; a++)
          for (int b = 0; b < 2; b++) B[a * 2 + b] = Nx[a][0] * Mx[0][b] + Nx[a][1] * Mx[1][b] + Nx[a][2] * Mx[2][b];
        rows_mul(L_, P_, s2_idx, 2, B, n);
        rows_mul(K_x, K_x, s2_idx, 2, B, C);
        cols_mulT(L_, s2_idx, 2, B, n);
        cols_mulT(P_, s2_idx, 2, B, n);
      }
      Mat KxL(n, C), Pt(C, n);  // :714
      for (int a = 0; a < n; a++)
        for (int b = 0; b < C; b++) KxL(a, b) = K_x(a, b);
      for (int a = 0; a < C; a++)
        for (int b = 0; b < n; b++) Pt(a, b) = P_(a, b);
      P_ = L_ - KxL * Pt;
      st.solve_time += omp_get_wtime() - solve_start;
      return;
    }
    st.solve_time += omp_get_wtime() - solve_start;
  }
}

}  // namespace orc

// ---------------------------------------------------------------------------------------------------------
// map_incremental(), laserMapping.cpp:398-446: which points of the scan go into the map, and through which
// Add_Points branch. Runs after the filter update: `state_point` is the posterior, Nearest_Points are those of
// the last search pass. feats_down_world[i].normal_y is whatever the caller's cloud held at index i (the
// reference never writes it on this path: 0.001 below the first scan's size, :1004, else PCL's default 0).
namespace orc {
static float calc_dist_pt(const Pt &a, const Pt &b) {  // ikd_Tree.cpp:1694-1699 via common_lib.h
  float dist = (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z);
  return dist;
}

void Scene::map_incremental(const State &state_point, bool flg_EKF_inited, std::vector<Pt> &PointToAdd,
                            std::vector<Pt> &PointNoNeedDownsample) {
  const int feats_down_size = (int)feats_down_body.size();
  const double filter_size_map_min = prm.filter_size_map;
  PointToAdd.clear(), PointNoNeedDownsample.clear();
  for (int i = 0; i < feats_down_size; i++) {
    if (feats_down_body[i].normal_y > prm.cov_threshold) continue;  // :406
    {  // pointBodyToWorld, :134-147
      const Pt *pi = &feats_down_body[i];
      Pt *po = &feats_down_world[i];
      V3 p_body{pi->x, pi->y, pi->z};
      V3 p_global;
      int lid_idx = (int)pi->intensity;
      if (lid_idx == 0)
        p_global = state_point.rot * (state_point.offset_R[lid_idx] * p_body + state_point.offset_T[lid_idx]) + state_point.pos;
      else
        p_global = state_point.rot * (temporal_comp[lid_idx - 1].q_ * (state_point.offset_R[lid_idx] * p_body + state_point.offset_T[lid_idx]) +
                                      temporal_comp[lid_idx - 1].t_) +
                   state_point.pos;
      po->x = (float)p_global.x, po->y = (float)p_global.y, po->z = (float)p_global.z;
      po->intensity = pi->intensity;
    }
    if (!Nearest_Points[i].empty() && flg_EKF_inited) {  // :411
      const std::vector<Pt> &points_near = Nearest_Points[i];
      bool need_add = true;
      Pt mid_point;
      mid_point.x = (float)(std::floor(feats_down_world[i].x / filter_size_map_min) * filter_size_map_min + 0.5 * filter_size_map_min);
      mid_point.y = (float)(std::floor(feats_down_world[i].y / filter_size_map_min) * filter_size_map_min + 0.5 * filter_size_map_min);
      mid_point.z = (float)(std::floor(feats_down_world[i].z / filter_size_map_min) * filter_size_map_min + 0.5 * filter_size_map_min);
      float dist = calc_dist_pt(feats_down_world[i], mid_point);
      if (std::fabs(points_near[0].x - mid_point.x) > 0.5 * filter_size_map_min &&
          std::fabs(points_near[0].y - mid_point.y) > 0.5 * filter_size_map_min &&
          std::fabs(points_near[0].z - mid_point.z) > 0.5 * filter_size_map_min) {  // :421-425
        PointNoNeedDownsample.push_back(feats_down_world[i]);
        continue;
      }
      for (int readd_i = 0; readd_i < NUM_MATCH_POINTS; readd_i++) {  // :426-435
        if ((int)points_near.size() < NUM_MATCH_POINTS) break;
        if (calc_dist_pt(points_near[readd_i], mid_point) < dist) {
          need_add = false;
          break;
        }
      }
      if (need_add) PointToAdd.push_back(feats_down_world[i]);
    } else {
      PointToAdd.push_back(feats_down_world[i]);
    }
  }
}
}  // namespace orc

// ---------------------------------------------------------------------------------------------------------
// esekf::predict / predict_cont / back_predict (esekfom.hpp:171-279, 281-385, 388-491) share one body: they differ
// only in WHICH state/covariance pair they advance (x_/P_, x_cont/P_unc_, x_unc/P_unc_). That body, with the process
// model of src/use-ikfom.hpp:67-112 (get_f, df_dx, df_dw) for the runtime-parametric state:
//   f_   (m = 18 + 6 L, "flatted": the S2 entry has 3 rows)     f_x_ (m x n), f_w_ (m x 12)
//   x.oplus(f_, dt)                                               (:398)
//   F_x1 = I; vect rows copied; SO3 rows: F_x1 block = exp(seg, scalar(1/2)) - integer division, i.e. exp with scale
//   0 = IDENTITY (the same quirk as S2.hpp:287) - and rows of f_x/f_w multiplied by A_matrix(seg), seg = -f dt;
//   S2 rows: F_x1 block = Nx(x) * exp(...)=I * Mx(x_before, 0), rows = -Nx * hat(x_before) * A_matrix(seg)^T * rows
//   F_x1 += f_x_final dt;  P = F_x1 P F_x1^T + (dt f_w_final) Q (dt f_w_final)^T          (:487-488)
// Process noise order (use-ikfom.hpp:29-35): ng, na, nbg, nba (3 each).
namespace orc {
void predict(State &x_, Mat &P_, double dt, const Mat &Q, V3 acc, V3 gyro) {
  const int L = x_.L, n = x_.dof(), m = n + 1;
  const int i_rot = 3, i_vel = 6 * (L + 1), i_bg = i_vel + 3, i_ba = i_vel + 6, i_grav = i_vel + 9;
  // get_f (:67-81)
  std::vector<double> f_(m, 0.0);
  V3 omega = gyro - x_.bg;
  V3 a_inertial = x_.rot * (acc - x_.ba);
  for (int i = 0; i < 3; i++) {
    f_[i] = x_.vel[i];
    f_[i + 3] = omega[i];
    f_[i + i_vel] = a_inertial[i] + x_.grav[i];
  }
  // df_dx (:83-101)
  Mat f_x_(m, n), f_w_(m, 12);
  M3 R = toR(x_.rot);
  V3 acc_ = acc - x_.ba;
  M3 RhA = (-1.0) * (R * hat(acc_));
  double grav_matrix[3][2];
  S2_Mx(x_.grav, 0.0, 0.0, grav_matrix);
  for (int i = 0; i < 3; i++) {
    f_x_(i, i_vel + i) = 1.0;
    f_x_(i_rot + i, i_bg + i) = -1.0;
    for (int j = 0; j < 3; j++) {
      f_x_(i_vel + i, i_rot + j) = RhA.m[i][j];
      f_x_(i_vel + i, i_ba + j) = -R.m[i][j];
    }
    for (int j = 0; j < 2; j++) f_x_(i_vel + i, i_grav + j) = grav_matrix[i][j];
  }
  // df_dw (:104-112)
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) f_w_(i_vel + i, 3 + j) = -R.m[i][j];
    f_w_(i_rot + i, i) = -1.0;
    f_w_(i_bg + i, 6 + i) = 1.0;
    f_w_(i_ba + i, 9 + i) = 1.0;
  }
  State x_before = x_;
  // x_.oplus(f_, dt): vect += f dt; SO3 *= exp(f dt) (scale/2 is a double division there, SOn.hpp:252); S2: rotate by
  // exp(f dt) (S2.hpp:129-134) - f is zero for the gravity entry, the extrinsics and the biases
  auto f3 = [&](int o) { return V3{f_[o], f_[o + 1], f_[o + 2]}; };
  x_.pos = x_.pos + dt * f3(0);
  x_.rot = x_.rot * so3_exp(f3(i_rot), dt);
  for (int l = 0; l < L; l++) x_.offset_R[l] = x_.offset_R[l] * so3_exp(f3(6 + 3 * l), dt);
  for (int l = 0; l < L; l++) x_.offset_T[l] = x_.offset_T[l] + dt * f3(6 + 3 * L + 3 * l);
  x_.vel = x_.vel + dt * f3(i_vel);
  x_.bg = x_.bg + dt * f3(i_bg);
  x_.ba = x_.ba + dt * f3(i_ba);
  x_.grav = toR(so3_exp(f3(i_grav), dt)) * x_.grav;

  Mat F_x1 = Mat::I(n), f_x_final(n, n), f_w_final(n, 12);
  // vect states: idx == dim for every block before the S2 entry
  auto copy_rows = [&](int idx, int dof) {
    for (int j = 0; j < dof; j++) {
      for (int i = 0; i < n; i++) f_x_final(idx + j, i) = f_x_(idx + j, i);
      for (int i = 0; i < 12; i++) f_w_final(idx + j, i) = f_w_(idx + j, i);
    }
  };
  copy_rows(0, 3);
  for (int l = 0; l < L; l++) copy_rows(6 + 3 * L + 3 * l, 3);
  copy_rows(i_vel, 3), copy_rows(i_bg, 3), copy_rows(i_ba, 3);
  // SO3 states
  std::vector<int> so3 = {i_rot};
  for (int l = 0; l < L; l++) so3.push_back(6 + 3 * l);
  for (int idx : so3) {
    V3 seg{-1 * f_[idx] * dt, -1 * f_[idx + 1] * dt, -1 * f_[idx + 2] * dt};
    M3 Rres = toR(mtk_exp_scale(seg, 0.0));  // scalar(1/2) == 0
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) F_x1(idx + i, idx + j) = Rres.m[i][j];
    M3 A = A_matrix(seg);
    for (int c = 0; c < n; c++) {
      V3 col{f_x_(idx, c), f_x_(idx + 1, c), f_x_(idx + 2, c)};
      V3 r = A * col;
      for (int k = 0; k < 3; k++) f_x_final(idx + k, c) = r[k];
    }
    for (int c = 0; c < 12; c++) {
      V3 col{f_w_(idx, c), f_w_(idx + 1, c), f_w_(idx + 2, c)};
      V3 r = A * col;
      for (int k = 0; k < 3; k++) f_w_final(idx + k, c) = r[k];
    }
  }
  {  // S2 state: idx = dim = i_grav
    V3 seg{f_[i_grav] * dt, f_[i_grav + 1] * dt, f_[i_grav + 2] * dt};
    M3 Rres = toR(mtk_exp_scale(seg, 0.0));
    double Nx[2][3], Mx[3][2];
    S2_Nx_yy(x_.grav, Nx);
    S2_Mx(x_before.grav, 0.0, 0.0, Mx);
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++) {
        double s = 0;
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) s += Nx[i][a] * Rres.m[a][b] * Mx[b][j];
        F_x1(i_grav + i, i_grav + j) = s;
      }
    M3 T = Rres * hat(x_before.grav) * transpose(A_matrix(seg));
    for (int c = 0; c < n; c++)
      for (int i = 0; i < 2; i++) {
        double s = 0;
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) s += -Nx[i][a] * T.m[a][b] * f_x_(i_grav + b, c);
        f_x_final(i_grav + i, c) = s;
      }
    for (int c = 0; c < 12; c++)
      for (int i = 0; i < 2; i++) {
        double s = 0;
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) s += -Nx[i][a] * T.m[a][b] * f_w_(i_grav + b, c);
        f_w_final(i_grav + i, c) = s;
      }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) F_x1(i, j) += f_x_final(i, j) * dt;
  Mat G(n, 12);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 12; j++) G(i, j) = dt * f_w_final(i, j);
  Mat Ft(n, n), Gt(12, n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Ft(j, i) = F_x1(i, j);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 12; j++) Gt(j, i) = G(i, j);
  Mat A1 = F_x1 * P_ * Ft, A2 = G * Q * Gt;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) P_(i, j) = A1(i, j) + A2(i, j);
}
}  // namespace orc
