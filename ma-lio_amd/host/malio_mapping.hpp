// Header-only C++ mirror of the reference's interface for the hot path, on top of the C ABI (include/malio.h).
// Same names, argument meaning and error behaviour as the reference so that laserMapping.cpp changes by a few
// lines. It deliberately depends on neither Eigen nor PCL: the caller's own types are
// layout-compatible (pcl::PointXYZINormal == malio_point_t, Pose == malio_pose_t up to Eigen's storage).
//
//   KD_TREE<PointType> ikdtree                    -> malio::KdTreeGpu            (laserMapping.cpp:95)
//   h_share_model(state_ikfom&, dyn_share&)       -> malio::Mapping::h_share_model (laserMapping.cpp:552)
//   kf.update_iterated_dyn_share_modified(R, t)   -> malio::Mapping::update_iterated_dyn_share_modified
//                                                                             (esekfom.hpp:495)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/malio.h"

namespace malio {

using PointType = malio_point_t;            // common_lib.h:31
using PointVector = std::vector<PointType>;  // common_lib.h:33
using BoxPointType = malio_box_t;           // ikd_Tree.h:25-29
using Pose = malio_pose_t;                  // common_lib.h:57-63

// esekfom::dyn_share_datastruct<double> (esekfom.hpp:80-90) without Eigen: row-major h_x (M x C).
struct dyn_share_datastruct {
  bool valid = true;
  bool converge = true;
  int rows = 0, cols = 0;
  std::vector<double> h_x, h, R;
  // fused form (fast path): what esekfom.hpp:621-635 computes from h_x / h / R
  std::vector<double> HtRinvH, HtRinvh;
  double w_loc = 0;
};

// One GPU (malio_create) or all GPUs of the node behind one handle (malio_node_create, include/malio.h): the classes below
// call malio_xxx or malio_node_xxx accordingly, so a loop written against them runs on either. The environment can
// turn a plain Handle into a node without touching the caller: MALIO_NODE="<gpus>[,scan|tiles|columns[,rccl]]" (every shard on
// `device` when MALIO_NODE_SAME_DEVICE=1: the single-GPU test boxes).
class Handle {
 public:
  Handle(const malio_params_t &prm, int device = 0) : prm_(prm) {
    if (const char *e = std::getenv("MALIO_NODE")) {
      const int g = std::atoi(e);
      if (g >= 1) {
        const bool tiles = std::strstr(e, "tiles") != nullptr, columns = std::strstr(e, "columns") != nullptr, rccl = std::strstr(e, "rccl") != nullptr;
        const char *same = std::getenv("MALIO_NODE_SAME_DEVICE");
        std::vector<int> dev(g);
        for (int r = 0; r < g; r++) dev[r] = (same && same[0] == '1') ? device : r;
        init_node(g, dev.data(), columns ? MALIO_PART_COLUMNS : tiles ? MALIO_PART_TILES : MALIO_PART_SCAN,
                  rccl ? MALIO_NODE_XCHG_RCCL : MALIO_NODE_XCHG_HOST, 0.f);
        return;
      }
    }
    int rc = malio_create(&prm, device, &h_);
    if (rc != MALIO_OK) throw std::runtime_error("malio_create failed (" + std::to_string(rc) + "): no gfx950 device?");
  }
  Handle(const malio_params_t &prm, int n_gpus, const int *devices, int partition, int exchange, float tile_m) : prm_(prm) {
    init_node(n_gpus, devices, partition, exchange, tile_m);
  }
  ~Handle() {
    if (h_) malio_destroy(h_);
    if (nd_) malio_node_destroy(nd_);
  }
  Handle(const Handle &) = delete;
  Handle &operator=(const Handle &) = delete;
  malio_handle_t get() const { return h_; }
  malio_node_t node() const { return nd_; }
  bool is_node() const { return nd_ != nullptr; }
  const malio_params_t &params() const { return prm_; }
  void check(int rc, const char *what) const {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + (nd_ ? malio_node_last_error(nd_) : malio_last_error(h_)));
  }

 private:
  void init_node(int n_gpus, const int *devices, int partition, int exchange, float tile_m) {
    int rc = malio_node_create(&prm_, n_gpus, devices, partition, exchange, tile_m, &nd_);
    if (rc != MALIO_OK) throw std::runtime_error("malio_node_create failed (" + std::to_string(rc) + ")");
  }
  malio_handle_t h_ = nullptr;
  malio_node_t nd_ = nullptr;
  malio_params_t prm_;
};

class KdTreeGpu {
 public:
  explicit KdTreeGpu(Handle &h) : h_(h) {}
  void set_downsample_param(float) {}                                  // laserMapping.cpp:999 (kept in params)
  void Build(const PointVector &pts) {                                 // :1007
    h_.check(h_.is_node() ? malio_node_map_build(h_.node(), pts.data(), (int)pts.size())
                          : malio_map_build(h_.get(), pts.data(), (int)pts.size()), "Build");
    built_ = true;
  }
  bool empty() const { return !built_; }                               // `Root_Node == nullptr` test, :995
  int size() const {                                                   // :824
    int n = 0;
    if (h_.is_node())
      malio_node_map_total(h_.node(), &n);
    else
      malio_map_size(h_.get(), &n);
    return n;
  }
  // Batched form of Nearest_Search (:586): all queries of a scan in one call.
  void Nearest_Search(const PointVector &queries, int k, std::vector<PointVector> &Nearest_Points,
                      std::vector<std::vector<float>> &Point_Distance) const {
    const int n = (int)queries.size();
    std::vector<PointType> out((size_t)n * k);
    std::vector<float> d2((size_t)n * k);
    std::vector<int> cnt(n);
    h_.check(h_.is_node() ? malio_node_nearest_search(h_.node(), queries.data(), n, k, out.data(), d2.data(), cnt.data())
                          : malio_nearest_search(h_.get(), queries.data(), n, k, out.data(), d2.data(), cnt.data()), "Nearest_Search");
    Nearest_Points.assign(n, PointVector());
    Point_Distance.assign(n, std::vector<float>());
    for (int i = 0; i < n; i++) {
      Nearest_Points[i].assign(out.begin() + (size_t)i * k, out.begin() + (size_t)i * k + cnt[i]);
      Point_Distance[i].assign(d2.begin() + (size_t)i * k, d2.begin() + (size_t)i * k + cnt[i]);
    }
  }
  int Add_Points(PointVector &PointToAdd, bool downsample_on) {         // :443-444
    int added = 0;
    if (h_.is_node()) {  // (every GPU reports its own count: a replica's is the reference's, a tile shard's its share)
      std::vector<int> per((size_t)malio_node_gpus(h_.node()), 0);
      h_.check(malio_node_map_add(h_.node(), PointToAdd.data(), (int)PointToAdd.size(), downsample_on ? 1 : 0, per.data()), "Add_Points");
      return per[0];
    }
    h_.check(malio_map_add(h_.get(), PointToAdd.data(), (int)PointToAdd.size(), downsample_on ? 1 : 0, &added), "Add_Points");
    return added;
  }
  int Delete_Point_Boxes(std::vector<BoxPointType> &BoxPoints) {        // :223
    int del = 0;
    if (h_.is_node()) {
      std::vector<int> per((size_t)malio_node_gpus(h_.node()), 0);
      h_.check(malio_node_map_delete_boxes(h_.node(), BoxPoints.data(), (int)BoxPoints.size(), per.data()), "Delete_Point_Boxes");
      return per[0];
    }
    h_.check(malio_map_delete_boxes(h_.get(), BoxPoints.data(), (int)BoxPoints.size(), &del), "Delete_Point_Boxes");
    return del;
  }
  // ikdtree.flatten(ikdtree.Root_Node, ikdtree.PCL_Storage, NOT_RECORD)  :1018-1019 (valid points, map order)
  void flatten(PointVector &Storage) const {
    int n = size();
    Storage.resize((size_t)n);
    h_.check(h_.is_node() ? malio_node_map_get(h_.node(), Storage.data(), n, &n) : malio_map_get(h_.get(), Storage.data(), n, &n), "flatten");
  }

 private:
  Handle &h_;
  bool built_ = false;
};

// pcl::VoxelGrid<PointType> downSizeFilterSurf (laserMapping.cpp:93): setLeafSize (:860), setInputCloud + filter (:970-971)
class VoxelGridGpu {
 public:
  explicit VoxelGridGpu(Handle &h) : h_(h) {}
  void setLeafSize(float lx, float, float) { leaf_ = lx; }  // the reference uses one size for the three axes
  void setInputCloud(const PointVector &cloud) { in_ = &cloud; }
  void filter(PointVector &output) {
    const int n = in_ ? (int)in_->size() : 0;
    output.resize((size_t)n);
    int m = 0;
    h_.check(h_.is_node() ? malio_node_voxel_downsample(h_.node(), n ? in_->data() : nullptr, n, leaf_, MALIO_VOXEL_NORMAL_NORMALIZE,
                                                        output.data(), n, &m)
                          : malio_voxel_downsample(h_.get(), n ? in_->data() : nullptr, n, leaf_, MALIO_VOXEL_NORMAL_NORMALIZE,
                                                   output.data(), n, &m), "VoxelGrid::filter");
    output.resize((size_t)m);
  }

 private:
  Handle &h_;
  const PointVector *in_ = nullptr;
  float leaf_ = 0.5f;
};

// The per-scan globals + the two calls on the hot path.
class Mapping {
 public:
  explicit Mapping(Handle &h) : h_(h) {}
  // feats_down_body (laserMapping.cpp:86,982), pose_unc (:1028-1048), kf.temporal_comp (IMU_Processing.hpp:510-522)
  void set_scan(const PointVector &feats_down_body, const std::vector<std::vector<Pose>> &pose_unc,
                const std::vector<Pose> &temporal_comp) {
    const int L = h_.params().lid_num;
    std::vector<const Pose *> ptr(L);
    std::vector<int> len(L);
    for (int l = 0; l < L; l++) ptr[l] = pose_unc[l].data(), len[l] = (int)pose_unc[l].size();
    feats_down_size_ = (int)feats_down_body.size();
    h_.check(h_.is_node() ? malio_node_scan_set(h_.node(), feats_down_body.data(), feats_down_size_, ptr.data(), len.data(),
                                                L > 1 ? temporal_comp.data() : nullptr)
                          : malio_scan_set(h_.get(), feats_down_body.data(), feats_down_size_, ptr.data(), len.data(),
                                           L > 1 ? temporal_comp.data() : nullptr), "set_scan");
  }
  // Resident front end: the undistorted clouds stay in HBM (one malio_undistort_resident call per LiDAR, see
  // INTEGRATION.md §4b), then downSizeFilterSurf + the field shuffle + the concatenation of laserMapping.cpp:966-983 run
  // on the device and the result becomes the scan. feats_down_body comes back for the caller's own bookkeeping.
  void set_scan_resident(float filter_size_surf, const std::vector<std::vector<Pose>> &pose_unc,
                         const std::vector<Pose> &temporal_comp, PointVector &feats_down_body, int max_points) {
    const int L = h_.params().lid_num;
    std::vector<const Pose *> ptr(L);
    std::vector<int> len(L);
    for (int l = 0; l < L; l++) ptr[l] = pose_unc[l].data(), len[l] = (int)pose_unc[l].size();
    feats_down_body.resize((size_t)max_points);
    int n = 0;
    h_.check(h_.is_node() ? malio_node_scan_set_resident(h_.node(), filter_size_surf, MALIO_VOXEL_NORMAL_NORMALIZE, ptr.data(), len.data(),
                                                         L > 1 ? temporal_comp.data() : nullptr, feats_down_body.data(), max_points, &n)
                          : malio_scan_set_resident(h_.get(), filter_size_surf, MALIO_VOXEL_NORMAL_NORMALIZE, ptr.data(), len.data(),
                                                    L > 1 ? temporal_comp.data() : nullptr, feats_down_body.data(), max_points, &n),
             "set_scan_resident");
    feats_down_body.resize((size_t)std::min(n, max_points));
    feats_down_size_ = n;
  }
  // void h_share_model(state_ikfom &s, esekfom::dyn_share_datastruct<double> &ekfom_data), laserMapping.cpp:552.
  // want_rows = true reproduces h_x / h / R exactly as the reference fills them (:642-644); false hands the
  // filter the reduced normal equations instead.
  void h_share_model(const malio_state_t &s, dyn_share_datastruct &ekfom_data, bool want_rows = false) {
    const int C = 6 * (1 + h_.params().lid_num);
    malio_measure_out_t out;
    std::memset(&out, 0, sizeof(out));
    const bool rows_on_node = want_rows && h_.is_node();
    if (h_.is_node()) want_rows = false;  // (the dense rows are a single-GPU path: the node hands out the normal equations;
                                          //  h_x / h / R come back zero-filled at their sizes, so that indexing stays valid)
    if (want_rows) {
      ekfom_data.h_x.assign((size_t)feats_down_size_ * C, 0.0);
      ekfom_data.h.assign(feats_down_size_, 0.0);
      ekfom_data.R.assign(feats_down_size_, 0.0);
      out.h_x = ekfom_data.h_x.data(), out.h = ekfom_data.h.data(), out.R = ekfom_data.R.data();
    }
    int rc = h_.is_node() ? malio_node_measure(h_.node(), &s, ekfom_data.converge ? 1 : 0, &out)
                          : malio_measure(h_.get(), &s, ekfom_data.converge ? 1 : 0, &out);
    h_.check(rc, "h_share_model");
    if (!out.valid) {  // laserMapping.cpp:635-639: ekfom_data.valid = false; ROS_WARN("No Effective Points!")
      ekfom_data.valid = false;
      return;
    }
    ekfom_data.rows = out.M, ekfom_data.cols = C, ekfom_data.w_loc = out.w_loc;
    ekfom_data.HtRinvH.assign(out.HtRinvH, out.HtRinvH + C * C);
    ekfom_data.HtRinvh.assign(out.HtRinvh, out.HtRinvh + C);
    if (want_rows) {
      ekfom_data.h_x.resize((size_t)out.M * C);
      ekfom_data.h.resize(out.M);
      ekfom_data.R.resize(out.M);
    }
    if (rows_on_node) {
      ekfom_data.h_x.assign((size_t)out.M * C, 0.0);
      ekfom_data.h.assign(out.M, 0.0);
      ekfom_data.R.assign(out.M, 0.0);
    }
  }
  // h_share_model of a scan sharded over the ranks of one node (malio_measure_node): same fused outputs, scan-global
  // weights, normal equations of the whole scan; every rank then runs the same filter step on the same bits
  void h_share_model_node(malio_xchg_t x, const malio_state_t &s, dyn_share_datastruct &ekfom_data) {
    const int C = 6 * (1 + h_.params().lid_num);
    malio_measure_out_t out;
    memset(&out, 0, sizeof(out));
    int rc = malio_measure_node(h_.get(), x, &s, ekfom_data.converge ? 1 : 0, &out, nullptr);
    h_.check(rc, "measure_node");
    ekfom_data.valid = out.valid != 0;
    ekfom_data.rows = out.M, ekfom_data.cols = C, ekfom_data.w_loc = out.w_loc;
    ekfom_data.HtRinvH.assign(out.HtRinvH, out.HtRinvH + (size_t)C * C);
    ekfom_data.HtRinvh.assign(out.HtRinvh, out.HtRinvh + C);
  }
  // the iterated update over a scan sharded across the ranks of one node (every rank: same x, P in, same posterior out)
  void update_iterated_dyn_share_modified_node(malio_xchg_t x, malio_state_t &state, std::vector<double> &P, double R,
                                               double &solve_time) {
    h_.check(malio_update_iterated_node(h_.get(), x, &state, P.data(), R, nullptr, &solve_time), "update_iterated_node");
  }
  // kf.update_iterated_dyn_share_modified(LASER_POINT_COV, solve_H_time), laserMapping.cpp:1052.
  // P: n x n row-major, n = 17 + 6 lid_num.
  void update_iterated_dyn_share_modified(malio_state_t &x, std::vector<double> &P, double R, double &solve_time) {
    h_.check(h_.is_node() ? malio_node_update_iterated(h_.node(), &x, P.data(), R, nullptr, &solve_time)
                          : malio_update_iterated(h_.get(), &x, P.data(), R, nullptr, &solve_time), "update_iterated");
  }
  // order of the scan inside the engine (malio_scan_order): MALIO_SCAN_ORDER_AUTO / _SORT / _KEEP
  void set_scan_order(int mode) {
    if (!h_.is_node()) h_.check(malio_scan_order(h_.get(), mode), "scan_order");
  }
  // kf.predict(dt, Q, in) (esekfom.hpp:388-492; also predict_cont :171 / back_predict :281 when handed x_cont /
  // x_unc and P_unc_), IMU_Processing.hpp:332,345,364,386,399. Host code; Q is 12 x 12 row-major.
  static void predict(int lid_num, malio_state_t &x, std::vector<double> &P, double dt, const std::vector<double> &Q,
                      const double acc[3], const double gyro[3]) {
    int rc = malio_predict(lid_num, &x, P.data(), dt, Q.data(), acc, gyro);
    if (rc != MALIO_OK) throw std::runtime_error("malio_predict rc=" + std::to_string(rc));
  }
  // "Calculate each LiDAR state uncertainty", laserMapping.cpp:1028-1048: the per-LiDAR tables h_share_model indexes,
  // from the backward-propagated kf.lidar_uncertainty (IMU_Processing.hpp:366-400). The last entry of every list is
  // dropped (`size() - 1`); secondary LiDARs are carried into the primary's frame by ext_l (+) entry, then the
  // temporal compensation, then ext_0^-1, each call writing over its own second argument as the reference does.
  static std::vector<std::vector<Pose>> pose_uncertainty_tables(const std::vector<Pose> &extrinsic,
                                                                const std::vector<std::vector<Pose>> &lidar_uncertainty,
                                                                const std::vector<Pose> &temporal_comp) {
    const int lid_num = (int)lidar_uncertainty.size();
    std::vector<std::vector<Pose>> pose_unc(lid_num);
    Pose pose_point{};
    for (int num = 0; num < lid_num; num++) {
      const int cnt = (int)lidar_uncertainty[num].size() - 1;
      for (int i = 0; i < cnt; i++) {
        if (num == 0) {
          pose_unc[num].push_back(lidar_uncertainty[num][i]);
          continue;
        }
        malio_compound_pose_cov(&extrinsic[num], &lidar_uncertainty[num][i], &pose_point);
        malio_compound_pose_cov(&temporal_comp[num - 1], &pose_point, &pose_point);
        malio_compound_inv_pose_cov(&extrinsic[0], &pose_point, &pose_point);
        pose_unc[num].push_back(pose_point);
      }
    }
    return pose_unc;
  }
  // void map_incremental(), laserMapping.cpp:398-446, without bringing Nearest_Points to the host.
  // state_point = kf.get_x() after the update (:1053); feats_down_world is the caller's cloud: only its normal_y
  // is read (the value the reference would store with each added point). Returns add_point_size (:445).
  int map_incremental(const malio_state_t &state_point, bool flg_EKF_inited, const PointVector &feats_down_world) {
    std::vector<float> wny(feats_down_size_, 0.f);
    for (int i = 0; i < feats_down_size_ && i < (int)feats_down_world.size(); i++) wny[i] = feats_down_world[i].normal_y;
    int counts[3] = {0, 0, 0};
    h_.check(h_.is_node() ? malio_node_map_incremental(h_.node(), &state_point, flg_EKF_inited ? 1 : 0, wny.data(), counts)
                          : malio_map_incremental(h_.get(), &state_point, flg_EKF_inited ? 1 : 0, wny.data(), counts), "map_incremental");
    return counts[0] + counts[1];
  }
  // the same side effects on the host, for a caller that keeps its own map_incremental (laserMapping.cpp:406,411-435).
  // Note: Nearest_Points here are the neighbours with d2 <= 5 (the plane-fit gate, :587), not the unbounded 5-NN.
  void get_side_effects(std::vector<float> &normal_y, std::vector<PointType> &Nearest_Points_flat,
                        std::vector<int> &nearest_count, std::vector<uint8_t> &point_selected_surf) {
    normal_y.resize(feats_down_size_);
    Nearest_Points_flat.resize((size_t)feats_down_size_ * 5);
    nearest_count.resize(feats_down_size_);
    point_selected_surf.resize(feats_down_size_);
    h_.check(h_.is_node() ? malio_node_scan_get(h_.node(), normal_y.data(), Nearest_Points_flat.data(), nearest_count.data(),
                                                point_selected_surf.data(), nullptr, nullptr, nullptr)
                          : malio_scan_get(h_.get(), normal_y.data(), Nearest_Points_flat.data(), nearest_count.data(),
                                           point_selected_surf.data(), nullptr, nullptr, nullptr), "scan_get");
  }

 private:
  Handle &h_;
  int feats_down_size_ = 0;
};

}  // namespace malio
