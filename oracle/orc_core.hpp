// TEST INFRASTRUCTURE — CPU oracle (see orc_math.hpp header). Restatement of the reference's
// per-scan measurement update, each function citing the /root/reference file:line it follows.
// PARITY UNPINNED versus the real Eigen/PCL build: the reference ships no tests, golden vectors
// or fixtures; the k-NN part IS pinned against the reference's own ikd-Tree
// compiled from source (oracle/ref_ikdtree -> oracle/_ref/libikd_ref.so).
#pragma once
#include <array>
#include "orc_math.hpp"

namespace orc {

// pcl::PointXYZINormal memory layout (48 B), field overloading per SURVEY.md §2.2
struct Pt {
  float x = 0, y = 0, z = 0, p0 = 1.f;
  float normal_x = 0, normal_y = 0, normal_z = 0, p1 = 0;
  float intensity = 0, curvature = 0, p2 = 0, p3 = 0;
};
static_assert(sizeof(Pt) == 48, "layout");

// common_lib.h:57-63
struct Pose {
  Q q_;
  V3 t_;
  double T_[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  double cov_[6][6] = {};
};

// parameters.cpp:17-65 / City.yaml / mapping_city.launch values
struct Params {
  int lid_num = 1;
  int max_iteration = 3;
  int extrinsic_est_en = 1;
  float plane_th = 0.4f;
  double cov_threshold = 0.5;
  double range_min = 0, range_max = 1;
  double point_cov_max = 0.00125, point_cov_min = 0.00075;
  double plane_cov_max = 1, plane_cov_min = 0.8;
  double localize_cov_max = 2, localize_cov_min = 0.3;
  double localize_thresh_max = 0.7, localize_thresh_min = 0.2;
  double filter_size_map = 0.5;
  double limit = 0.001;  // esekf::limit[i] (esekfom.hpp:160-163,894)
};

constexpr int MAXL = 4;
constexpr int NUM_MATCH_POINTS = 5;  // common_lib.h:22

// use-ikfom.hpp:14-27 (runtime-parametric in lid_num; tangent layout SURVEY.md §2.1)
struct State {
  int L = 1;
  V3 pos;
  Q rot;
  Q offset_R[MAXL];
  V3 offset_T[MAXL];
  V3 vel, bg, ba;
  V3 grav{0, 0, -9.809};  // S2<double, 98090, 10000, 1>: |grav| = 9.809
  int dof() const { return 17 + 6 * L; }
};

// ---------------------------------------------------------------------------------------
// k-NN provider. The reference calls ikdtree.Nearest_Search(point_world, 5, near, d2)
// (laserMapping.cpp:586). Two providers: the reference's own tree via oracle/_ref, or an
// independent exact k-d tree restating ikd_Tree.cpp:1073-1255 semantics (float32 squared
// distances computed as ikd_Tree.cpp:1697, ascending output as ikd_Tree.cpp:452-458).
struct Knn {
  virtual ~Knn() {}
  virtual void build(const std::vector<Pt> &pts) = 0;
  virtual void search(const Pt &q, int k, std::vector<Pt> &near, std::vector<float> &d2) = 0;
  virtual int size() = 0;
};
Knn *make_kd_knn();
Knn *make_ref_knn(const char *so_path, float downsample);  // nullptr if the .so is unavailable

// common_lib.h:144-190
bool esti_plane(float pca_result[4], const std::vector<Pt> &point, float threshold, double &plane_cov,
                double cov_threshold);
// associate_uct.hpp:153-175 (+ pointToFS :145-151)
void evalPointUncertainty(const Pt &pi, double cov_point[3][3], const Pose &pose);
// associate_uct.hpp:85-142 / :29-83
void compoundPoseWithCov(const Pose &pose_1, const double cov_1[6][6], const Pose &pose_2, const double cov_2[6][6],
                         Pose &pose_cp, double cov_cp[6][6]);
void compoundInvPoseWithCov(const Pose &pose_1, const double cov_1[6][6], const Pose &pose_2,
                            const double cov_2[6][6], Pose &pose_cp, double cov_cp[6][6]);
void PoseInitial(Pose &pose, V3 trans, Q quat, const double cov[6][6]);  // common_lib.h:129-142

// esekfom.hpp:80-90
struct DynShare {
  bool valid = true;
  bool converge = true;
  Mat h_x;                // M x 6(1+L)
  std::vector<double> h;  // M
  std::vector<double> R;  // M
};

// The globals h_share_model touches (laserMapping.cpp:55-56,61-67,84-95), as one object.
struct Scene {
  Params prm;
  Knn *knn = nullptr;
  int threads = 1;  // MP_PROC_NUM (CMakeLists.txt:18-36); reference ships 3
  std::vector<Pt> feats_down_body, feats_down_world, normvec, laserCloudOri, corr_normvect;
  std::vector<std::vector<Pt>> Nearest_Points;
  std::vector<char> point_selected_surf;  // bool[100000] in the reference; cap lifted
  std::vector<float> res_last;
  std::vector<double> cov_plane;
  std::vector<std::vector<Pose>> pose_unc;  // [lid][k], laserMapping.cpp:1028-1048
  std::vector<Pose> temporal_comp;          // kf.temporal_comp, IMU_Processing.hpp:510-522
  int effct_feat_num = 0;
  double last_weight = 0;  // localization weight actually applied (laserMapping.cpp:749-756)
  // Multi-GPU test support: the four scan-global extrema of :615-616,646-647 as this shard
  // computed them, and an optional override holding the all-reduced values. Not part of the reference.
  double last_minmax[4] = {0, 1000, 0, 9999};  // max_unit_cov, min_unit_cov, max_cov, min_cov (local)
  bool use_override = false;
  double override_minmax[4] = {0, 0, 0, 0};
  bool skip_loc_weight = false;  // leave h_x / h un-weighted by the localization weight (it is global too)
  // test support: called before every h_dyn_share invocation of update_iterated with the pass number (0-based); the
  // reference's hook is a plain function that may do anything, e.g. find its map changed under it
  // test support (oracle/ref_eigen pin): when non-empty, call k of h_share_model hands back replay[k] instead of
  // evaluating the scan - the same recorded rows are fed to the reference's esekf built against Eigen
  std::vector<DynShare> replay;
  int replay_pos = 0;
  void (*pass_hook)(int pass, void *user) = nullptr;
  void *pass_hook_user = nullptr;
  void set_scan(const std::vector<Pt> &body);
  // laserMapping.cpp:552-760
  void h_share_model(const State &s, DynShare &ekfom_data);
  // laserMapping.cpp:398-446 (selection only; the two Add_Points calls are the caller's)
  void map_incremental(const State &state_point, bool flg_EKF_inited, std::vector<Pt> &PointToAdd,
                       std::vector<Pt> &PointNoNeedDownsample);
};

// State manifold ops (build_manifold.hpp:193-201 -> vect.hpp, SOn.hpp:241-247, S2.hpp:136-167)
void boxplus(State &x, const std::vector<double> &dx);
void boxminus(const State &x, const State &other, std::vector<double> &res);

struct UpdateStats {
  int passes = 0, searches = 0, last_M = 0;
  double solve_time = 0;
};
// esekfom.hpp:495-721 (update_iterated_dyn_share_modified). P is n x n, in/out.
void update_iterated(Scene &sc, State &x, Mat &P, double R, UpdateStats &st,
                     std::vector<State> *trace_states = nullptr);

// esekfom.hpp:388-492 (predict; predict_cont :171-279 and back_predict :281-385 are the same body on other members)
// with the process model of use-ikfom.hpp:67-112. Q is 12 x 12 (ng, na, nbg, nba).
void predict(State &x, Mat &P, double dt, const Mat &Q, V3 acc, V3 gyro);

// ---- undistortion (a13/a14) ----------------------------------------------------------------
// BsplineSE3.cpp:26-118,121-230 ; quat_ops.h:87-92,151-257
struct Spline {
  std::vector<std::pair<double, std::array<double, 16>>> control_points;  // sorted by time
  double dt = 0.01, timestamp_start = 0;
  void feed_trajectory(const std::vector<std::array<double, 8>> &traj_points);
  bool get_pose(double timestamp, Q &q_GtoI, V3 &p_IinG) const;
};

}  // namespace orc
