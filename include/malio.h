/*
 * malio.h — C ABI of libmalio_hip.so: the MI355X (gfx950) replacement for MA-LIO's per-scan
 * measurement-update hot path. The reference has no FFI layer; the path is reached
 * through ONE function-pointer hook and ONE class API, and every entry point below names the
 * reference interface it replaces (paths relative to /root/reference/MA_LIO):
 *
 *   hook : void h_share_model(state_ikfom&, esekfom::dyn_share_datastruct<double>&)
 *          src/laserMapping.cpp:552, registered at :852, invoked at
 *          include/IKFoM_toolkit/esekfom/esekfom.hpp:512
 *   class: KD_TREE<PointType> ikdtree   src/laserMapping.cpp:95, include/ikd-Tree/ikd_Tree.h:308-340
 *
 * Conventions: extern "C", opaque handle, plain pointers + sizes, caller-owned HOST buffers unless a
 * parameter is named d_* (device pointer). Every function returns an int status: 0 = MALIO_OK,
 * > 0 = soft condition (e.g. MALIO_NO_EFFECTIVE_POINTS), < 0 = hard error; no exceptions cross the
 * boundary. One handle = one GPU + one HIP stream; calls on a handle are not re-entrant (the
 * reference calls from its single main thread). There is NO CPU fallback: if no gfx950 device is
 * present malio_create fails with MALIO_ERR_NO_DEVICE.
 *
 * INTEGRATION.md shows the shim a MA-LIO maintainer adds around these calls.
 */
#ifndef MALIO_H_
#define MALIO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MALIO_ABI_VERSION 2
#define MALIO_MAX_LIDAR 4       /* reference ships lid_num in {1,2,3} (src/use-ikfom.hpp:12-41) */
#define MALIO_NUM_MATCH_POINTS 5 /* include/common_lib.h:22 */

enum {
  MALIO_OK = 0,
  MALIO_NO_EFFECTIVE_POINTS = 1, /* ekfom_data.valid = false, src/laserMapping.cpp:635-639 */
  MALIO_SMALL_M_FALLBACK = 2,    /* M < n: caller must use the rows path (esekfom.hpp:574-582) */
  MALIO_ERR_NO_DEVICE = -1,
  MALIO_ERR_HIP = -2,
  MALIO_ERR_BAD_ARG = -3,
  MALIO_ERR_NO_MAP = -4,
  MALIO_ERR_NO_SCAN = -5,
  MALIO_ERR_ALLOC = -6,
  MALIO_ERR_TIMEOUT = -7 /* malio_xchg_all_gather: a rank did not show up */
};

typedef struct malio_ctx *malio_handle_t;

/* == pcl::PointXYZINormal, 48 bytes (typedef PointType, include/common_lib.h:31). Field overloading
 * along the pipeline is tabulated in SURVEY.md §2.2: after the voxel filter `intensity` = LiDAR slot,
 * `normal_x` = mean uncertainty-interval index, `normal_y` = trace(Sigma_p) / map-point uncertainty,
 * `curvature` = time offset [ms]. A PointCloudXYZI's points.data() can be passed as is. */
typedef struct malio_point {
  float x, y, z, _pad0;
  float normal_x, normal_y, normal_z, _pad1;
  float intensity, curvature, _pad2, _pad3;
} malio_point_t;

/* == struct Pose, include/common_lib.h:57-63 (q_ as x,y,z,w; T_ and cov_ row-major). */
typedef struct malio_pose {
  double q[4];
  double t[3];
  double T[16];
  double cov[36];
} malio_pose_t;

/* == BoxPointType, include/ikd-Tree/ikd_Tree.h:25-29 */
typedef struct malio_box {
  float vertex_min[3];
  float vertex_max[3];
} malio_box_t;

/* The globals readParameters fills (src/parameters.cpp:17-65; values SURVEY.md §5.1). */
typedef struct malio_params {
  int32_t lid_num;          /* common/lid_num */
  int32_t max_iteration;    /* NUM_MAX_ITERATIONS */
  int32_t extrinsic_est_en; /* mapping/extrinsic_est_en */
  float plane_th;           /* plane_th */
  double cov_threshold;
  double range_min, range_max;
  double point_cov_max, point_cov_min;
  double plane_cov_max, plane_cov_min;
  double localize_cov_max, localize_cov_min;
  double localize_thresh_max, localize_thresh_min;
  double filter_size_map; /* filter_size_map_min: ikdtree.set_downsample_param, laserMapping.cpp:999 */
  float cell_size;        /* level-1 neighbour-list cell edge [m]; 0 = default 1.125 (>= sqrt(5)/2; level 2 uses twice this) */
  int32_t reserved;
  double limit;           /* convergence threshold of the iterated update on every tangent component: esekf's member
                           * `limit[n]` (esekfom.hpp:894), compared at :649-657; init_dyn_share sets every entry to
                           * 0.001 (:160-163). 0 = that value. A tighter limit keeps the loop iterating up to
                           * max_iteration + 1 passes (BASELINE config 5: "10 IESKF iterations"). */
} malio_params_t;

/* == the parts of state_ikfom (src/use-ikfom.hpp:14-27) h_share_model reads, plus the rest of the
 * manifold for malio_update_iterated. Quaternions are (x,y,z,w). offset_R/offset_T are the ITERATED
 * extrinsics (extrinsic_update() aliases the filter's live state, laserMapping.cpp:291-308). */
typedef struct malio_state {
  double pos[3];
  double rot[4];
  double offset_R[MALIO_MAX_LIDAR][4];
  double offset_T[MALIO_MAX_LIDAR][3];
  double vel[3], bg[3], ba[3];
  double grav[3]; /* S2<double,98090,10000,1>: |grav| = 9.809 */
} malio_state_t;

/* Result of one fused pass. With C = 6(1+lid_num):
 *   HtRinvH[C*C] (row-major, full symmetric) = H^T diag(1/R) H   (esekfom.hpp:622-629, R clamp :624-626)
 *   HtRinvh[C]                               = H^T diag(1/R) h   (the HT * dyn_share.h factor of :635)
 * both already include the localization weight w_loc (laserMapping.cpp:758-759 scales H and h by w,
 * hence w^2 here). Optional row outputs reproduce ekfom_data.h_x / h / R exactly as the reference fills
 * them (laserMapping.cpp:642-644,679-722,758-759), M rows in ascending scan index. */
typedef struct malio_measure_out {
  int32_t valid;         /* ekfom_data.valid */
  int32_t M;             /* effct_feat_num */
  double w_loc;          /* localization weight applied (laserMapping.cpp:749-756) */
  double unit_cov_minmax[2]; /* min/max_unit_cov (:615-616,625-628) */
  double R_minmax[2];        /* min_cov / max_cov (:646-647,700-703) */
  double HtRinvH[36 * (1 + MALIO_MAX_LIDAR) * (1 + MALIO_MAX_LIDAR)];
  double HtRinvh[6 * (1 + MALIO_MAX_LIDAR)];
  double *h_x; /* optional, capacity N*C, row-major M x C */
  double *h;   /* optional, capacity N */
  double *R;   /* optional, capacity N */
} malio_measure_out_t;

/* ---- lifetime ------------------------------------------------------------------------------ */
/* replaces: the file-scope globals + KD_TREE ctor (laserMapping.cpp:55-115). device = HIP ordinal. */
int malio_create(const malio_params_t *params, int device, malio_handle_t *out);
int malio_destroy(malio_handle_t h);
const char *malio_version(void);
/* 12 hex digits naming the sources + compiler flags this binary was built from (profiles/ record it, bench.py compares) */
const char *malio_build_id(void);
/* HIP devices this process can use (0: none - malio_create would return MALIO_ERR_NO_DEVICE). Asked of the library's own
 * HIP runtime: a caller that probes through a second copy of libamdhip64 (another search path, another version)
 * initialises a second runtime in the process, and the one that comes second finds no device. */
int malio_device_count(void);
const char *malio_last_error(malio_handle_t h);
/* external != 0: run on the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream; the value 0 /
 * NULL then means the legacy default stream, which is what PyTorch uses unless told otherwise).
 * external == 0: go back to the handle's own non-blocking stream. */
int malio_set_stream(malio_handle_t h, void *hip_stream, int external);

/* ---- map: KD_TREE<PointType> call sites ------------------------------------------------------ */
/* ikdtree.Build(feats_down_world->points)            laserMapping.cpp:1007 / ikd_Tree.cpp:369-397 */
int malio_map_build(malio_handle_t h, const malio_point_t *pts, int n);
/* ikdtree.size()                                     laserMapping.cpp:824 */
int malio_map_size(malio_handle_t h, int *out_size);
/* ikdtree.Nearest_Search(point, k, near, d2), batched laserMapping.cpp:586 / ikd_Tree.cpp:426-461.
 * Exact k-NN (k <= 5) inside radius 2*cell_size (>= sqrt(5) m, the reference's own acceptance gate
 * laserMapping.cpp:587); float32 squared distances computed as ikd_Tree.cpp:1697; ascending.
 * out_pts [n*k], out_d2 [n*k] (INFINITY-padded), out_count [n]. */
int malio_nearest_search(malio_handle_t h, const malio_point_t *queries, int n, int k, malio_point_t *out_pts,
                         float *out_d2, int *out_count);
/* ikdtree.Add_Points(PointToAdd, downsample_on)      laserMapping.cpp:443-444 / ikd_Tree.cpp:478-584.
 * downsample_on != 0: per voxel of edge params.filter_size_map (set_downsample_param, laserMapping.cpp:1005) the
 * reference's keeper rule is replayed in input order (one GPU thread per voxel, all voxels in parallel);
 * *out_added = the reference's return value (number of insertions performed; 0 on the no-downsample branch).
 * Map point indices change: read Nearest_Points (malio_scan_get) BEFORE calling this, as map_incremental does.
 * Only x, y, z, normal_y of a map point are stored (the fields the hot path and Add_Points read). */
int malio_map_add(malio_handle_t h, const malio_point_t *pts, int n, int downsample_on, int *out_added);
/* ikdtree.Delete_Point_Boxes(cub_needrm)             laserMapping.cpp:223 / ikd_Tree.cpp:643-669.
 * A point is inside a box iff vertex_min <= p < vertex_max on every axis (ikd_Tree.cpp:807,1263-1274).
 * *out_deleted = number of points removed. */
int malio_map_delete_boxes(malio_handle_t h, const malio_box_t *boxes, int nb, int *out_deleted);
/* map_incremental()                                   laserMapping.cpp:398-446, on the resident scan, no host copy
 * of Nearest_Points: skips points whose normal_y exceeds cov_threshold (:406), recomputes the world point with
 * `state_point` (the posterior; pointBodyToWorld :134-147), sorts every remaining point into PointToAdd or
 * PointNoNeedDownsample from the neighbours of the last search pass (:411-440), then performs
 * Add_Points(PointToAdd, true) and Add_Points(PointNoNeedDownsample, false) (:443-444).
 * world_normal_y: [N] in scan order, the value feats_down_world->points[i].normal_y holds on the caller's side (the
 * reference never writes it on this path; it is what ends up stored in the map), or NULL for zeros.
 * out_counts3 (may be NULL): |PointToAdd|, |PointNoNeedDownsample|, return value of the first Add_Points. */
int malio_map_incremental(malio_handle_t h, const malio_state_t *state_point, int flg_EKF_inited,
                          const float *world_normal_y, int *out_counts3);
/* The selection of map_incremental alone (laserMapping.cpp:398-442), the map untouched: out_pts[0, counts2[0]) =
 * PointToAdd, out_pts[counts2[0], counts2[0] + counts2[1]) = PointNoNeedDownsample, both in scan order, as (x, y, z,
 * normal_y); out_index (may be NULL): the scan index of each. On a map shard (malio_set_partition) the points this
 * shard serves. What malio_node_map_incremental merges over its GPUs before every shard is handed both lists. */
int malio_map_incremental_select(malio_handle_t h, const malio_state_t *state_point, int flg_EKF_inited,
                                 const float *world_normal_y, malio_point_t *out_pts, int *out_index, int cap,
                                 int *out_counts2);
/* ikdtree.flatten(Root_Node, PCL_Storage, NOT_RECORD) laserMapping.cpp:1018-1019 (map publishing / saving).
 * Copies min(cap, size) valid points in map order, *out_n = size. Order differs from the tree's traversal. */
int malio_map_get(malio_handle_t h, malio_point_t *out, int cap, int *out_n);

/* ---- sensor decode: City-dataset records -> the cloud Preprocess::process hands on ----- */
/* Livox Avia / Tele: 19-byte records x y z (f32 LE), reflectivity, tag, line (u8), offset_time (u32 LE)
 * (file_player/src/ROSThread.cpp:776-796,817-833) through Preprocess::avia_handler (MA_LIO/src/preprocess.cpp:59-107):
 * tag/line test, every point_filter_num-th valid point, curvature = offset_time / 1e6 [ms] (dropped above 100),
 * "differs from the previous point or outside the blind sphere" test with the precedence as written (:96).
 * eof_point != 0 reproduces file_player's `while(!file.eof())` loop, which appends one default (all-zero) point.
 * out: pl_surf in record order; *maximum_time as Preprocess::maximum_time (-9999 when nothing was looked at). */
int malio_decode_livox(malio_handle_t h, const unsigned char *records, int n_records, int n_scans, int point_filter_num,
                       double blind, int eof_point, malio_point_t *out, int cap, int *out_n, double *maximum_time);
/* Ouster: 22-byte records x y z intensity (f32 LE), ring (u16), t (u32 LE) (ROSThread.cpp:947-957) through
 * Preprocess::oust64_handler (preprocess.cpp:109-149): every point_filter_num-th record, blind sphere,
 * curvature = t * time_unit_scale * 1e-9f [ms]. */
int malio_decode_ouster(malio_handle_t h, const unsigned char *records, int n_records, int point_filter_num, double blind,
                        float time_unit_scale, malio_point_t *out, int cap, int *out_n, double *maximum_time);

/* Velodyne: the data[] of a sensor_msgs::PointCloud2 as the velodyne driver publishes it, through
 * Preprocess::velodyne_handler (preprocess.cpp:148-212). pcl::fromROSMsg (:155) maps the message's fields BY NAME onto
 * velodyne_ros::Point (preprocess.h:18-34: x, y, z, intensity, time - all FLOAT32 - and ring, UINT16), so the caller hands
 * over the message's point_step and the byte offsets of those fields (sensor_msgs::PointField::offset; little-endian, no
 * alignment assumed). off_intensity / off_time < 0: the message has no such field - the value reads 0.f. ring is not an
 * argument: the handler reads it only in the block of :161-186, whose results (given_offset_time, yaw_first, yaw_end)
 * nothing uses. Per point, as :189-211: curvature = time * time_unit_scale (float); every point_filter_num-th POINT (the
 * index counts all points, :202) that lies outside the blind sphere (x*x + y*y + z*z in float > blind * blind, :204) is
 * pushed; *maximum_time = the largest curvature pushed, or -9999 (:188,206-207; offsets may be negative: some drivers stamp
 * relative to the END of the sweep). n_points == 0: the handler returns before it touches maximum_time (:157-158) -
 * *out_n = 0 and *maximum_time is left as the caller had it.
 * CALLER'S SIDE of the contract (what pcl::fromROSMsg checks and this entry point cannot, having no field table): a field
 * whose PointField::datatype is not FLOAT32 (7) - a FLOAT64 or UINT32 `time`, say - is NOT mapped by fromROSMsg and reads
 * 0 in the reference: pass its offset as -1 (absent), never the offset of a field of another type (the bytes would be
 * reinterpreted). The message must be little-endian (is_bigendian == 0: fromROSMsg byte-swaps otherwise, this does not) and
 * its points contiguous (row_step == width * point_step; data[] = height * row_step bytes): hand a padded message over row
 * by row, or repack it. */
typedef struct malio_pc2_layout {
  int point_step;                                 /* bytes per point (sensor_msgs::PointCloud2::point_step) */
  int off_x, off_y, off_z, off_intensity, off_time; /* byte offset of each FLOAT32 field inside a point; < 0: absent */
} malio_pc2_layout_t;
int malio_decode_velodyne(malio_handle_t h, const unsigned char *data, int n_points, const malio_pc2_layout_t *layout,
                          int point_filter_num, double blind, float time_unit_scale, malio_point_t *out, int cap, int *out_n,
                          double *maximum_time);

/* ---- voxel down-sampling ------------------------------------------------ */
/* downSizeFilterSurf.setInputCloud(cloud); downSizeFilterSurf.filter(*out)   laserMapping.cpp:93,860,968-971:
 * pcl::VoxelGrid<PointType> with its defaults (all fields averaged, min_points_per_voxel 0), leaf = filter_size_surf.
 * Output: one point per occupied voxel in ascending voxel-index order (PCL's order); *out_n = number of voxels, at
 * most `cap` points are written. PCL is not vendored with the reference: its published algorithm is restated
 * (csrc/voxel.hip) and parity with a PCL build is unpinned. normal_mode selects what happens to the summed
 * normal_x/y/z: divided by the count, or normalised to unit length (pcl::CentroidPoint's AccumulatorNormal). */
#define MALIO_VOXEL_NORMAL_MEAN 0
#define MALIO_VOXEL_NORMAL_NORMALIZE 1
int malio_voxel_downsample(malio_handle_t h, const malio_point_t *pts, int n, float leaf, int normal_mode,
                           malio_point_t *out, int cap, int *out_n);

/* ---- resident front end: undistortion -> voxel filter -> scan, without host round trips of the clouds -------- */
/* malio_undistort_resident == malio_undistort for LiDAR `lid`, but the undistorted cloud stays in HBM; only the entry
 * point indices (and, if wanted, those few points) come back for the uncertainty tables (IMU_Processing.hpp:484-494).
 * malio_scan_set_resident then runs downSizeFilterSurf on every resident cloud (laserMapping.cpp:968-971), applies
 * normal_x <- intensity, intensity <- LiDAR number (:972-976), concatenates in LiDAR order (:982) and installs the
 * result as the scan (== malio_scan_set on that cloud). out_body (may be NULL) receives feats_down_body, *out_n its
 * size; scan indices used by malio_scan_get / malio_map_incremental are positions in that cloud.
 * out_entry_point / out_entry_pts (both optional) need room for n_imu entries. */
int malio_undistort_resident(malio_handle_t h, int lid, const malio_point_t *pts, int n, double lidar_beg_time,
                             const double *knot_times, const double *knot_poses, int n_knots, const double ext_q[4],
                             const double ext_t[3], const double end_q[4], const double end_t[3],
                             const double *imu_stamps, int n_imu, int cov_pointer0, int *out_entry_point,
                             int *out_n_entries, malio_point_t *out_entry_pts);
int malio_scan_set_resident(malio_handle_t h, float leaf, int normal_mode, const malio_pose_t *const *pose_unc,
                            const int *pose_unc_len, const malio_pose_t *temporal_comp, malio_point_t *out_body, int cap,
                            int *out_n);

/* ---- per scan ------------------------------------------------------------------------------- */
/* replaces the per-scan globals h_share_model reads: feats_down_body (laserMapping.cpp:86,982),
 * pose_unc[lid][k] (:1028-1048), kf.temporal_comp[lid-1] (IMU_Processing.hpp:510-522). Uploads the
 * scan to HBM once; resets Nearest_Points / point_selected_surf (:1024-1025).
 * LIFETIME OF feats_down_body. From ordinary (pageable) memory the cloud has been read completely when the call returns.
 * From page-locked memory (malio_host_alloc, hipHostMalloc, hipHostRegister - detected with hipPointerGetAttributes on the
 * first and the last byte) the call returns with ONE DMA copy of the n * 48 bytes still in flight (~90 us for 100 k
 * points): the caller must neither modify nor free the buffer until malio_scan_upload_wait(h) has returned, or any
 * later call on the handle that waits for its results has (malio_measure, malio_update_iterated, malio_scan_get, ...).
 * malio_host_free of the buffer is safe at any time (hipHostFree waits for the device). A caller that cannot promise
 * that sets MALIO_SCAN_SET_SYNC=1 in the environment (the call then waits for the copy) or keeps its cloud pageable. */
int malio_scan_set(malio_handle_t h, const malio_point_t *feats_down_body, int n,
                   const malio_pose_t *const *pose_unc, const int *pose_unc_len,
                   const malio_pose_t *temporal_comp);
/* The same scan handed over as 20-byte records instead of 48-byte points: what the engine keeps of a point. A caller
 * that already walks feats_down_body once per scan (laserMapping.cpp:972-976 rewrites normal_x and intensity of every
 * point right after the voxel filter) can fill these in that loop; the upload then moves 2.0 MB instead of 4.8 MB per
 * 100 k points (38 us instead of 90 us of PCIe time on the path to the first pass). w = LiDAR slot (int(intensity),
 * :570) in bits 0-7 | int(normal_x) (the uncertainty-table index of :694,737, clamped to +-0x3FFFFF) << 8. Indices of
 * malio_scan_get / malio_map_incremental are positions in this array. Lifetime of the buffer as for malio_scan_set. */
typedef struct malio_scan_rec {
  float x, y, z;   /* body frame */
  uint32_t w;      /* lid | (uint32_t)idx << 8 */
  float normal_y;  /* feats_down_body[i].normal_y as it comes in (returned untouched where the reference does not write it) */
} malio_scan_rec_t;
int malio_scan_set_packed(malio_handle_t h, const malio_scan_rec_t *recs, int n, const malio_pose_t *const *pose_unc,
                          const int *pose_unc_len, const malio_pose_t *temporal_comp);
/* Blocks until the upload queued by the last malio_scan_set has left the caller's buffer (an event recorded right
 * behind the copy: it does not wait for kernels queued after it). Returns at once when nothing is in flight. */
int malio_scan_upload_wait(malio_handle_t h);
/* The NEXT scan's cloud on its way to HBM while the current scan is still being worked on: `buf` = the page-locked array
 * a following malio_scan_set (packed == 0: n malio_point_t) or malio_scan_set_packed (packed != 0: n malio_scan_rec_t)
 * will be handed. The copy runs on a stream of its own into a spare device buffer - call it right before
 * malio_map_incremental and its 90 us (38 us packed, 100 k points) are hidden behind that call. The scan_set of the SAME
 * pointer and count then copies nothing; any other scan_set simply ignores what was staged. No counterpart in the
 * reference (its clouds never leave the host); feats_down_body of scan k+1 is ready at this point when the caller
 * undistorts scan k+1 (IMU_Processing.hpp:475-507 needs the posterior of scan k, not its map_incremental) before it
 * calls map_incremental for scan k. Lifetime of the buffer: untouched until malio_scan_upload_wait after that scan_set.
 * MALIO_ERR_BAD_ARG for a buffer that is not page-locked. */
int malio_scan_stage(malio_handle_t h, const void *buf, int n, int packed);

/* ONE h_share_model pass (laserMapping.cpp:552-760) fused with the H^T R^-1 H / H^T R^-1 h
 * accumulation of esekfom.hpp:621-635. converge = ekfom_data.converge (search vs neighbour reuse,
 * :583-591). Returns MALIO_OK or MALIO_NO_EFFECTIVE_POINTS. */
int malio_measure(malio_handle_t h, const malio_state_t *s, int converge, malio_measure_out_t *out);

/* Side effects later reference code relies on, in original scan order; any pointer
 * may be NULL: feats_down_body[i].normal_y (:699,730,741) | Nearest_Points[i] (:582, read by
 * map_incremental :411-435) + sizes | point_selected_surf[i] | res_last[i] (:609) |
 * feats_down_world xyz (:576-578) | normvec (n, pd2) (:604-607).
 * Nearest_Points are the 5 nearest map points of the last SEARCH pass at ANY distance, ascending, like
 * ikdtree.Nearest_Search leaves them (ikd_Tree.cpp:426-461 has no radius; nearest_count = min(5, map size)): points
 * with fewer than five neighbours inside the sqrt(5) m acceptance radius get an exact unrestricted search here, so
 * the reference's own map_incremental loop (:411-435) can run unchanged on what this returns. */
int malio_scan_get(malio_handle_t h, float *normal_y, malio_point_t *nearest, int *nearest_count,
                   uint8_t *selected, float *res_last, float *world_xyz, float *normvec4);

/* kf.update_iterated_dyn_share_modified(R, solve_time) (esekfom.hpp:495-721) with the fused measure
 * pass as h_dyn_share: state x (in/out), covariance P (n x n row-major, n = 17+6 lid_num, in/out).
 * stats (optional, int[4]) = {passes, searches, last M, converged-count t}. */
int malio_update_iterated(malio_handle_t h, malio_state_t *x, double *P, double R, int *stats,
                          double *solve_time);

/* How malio_update_iterated drives its loop (same arithmetic in all three).
 * MALIO_UPDATE_GATED (default): every pass of the loop is enqueued ahead of the GPU; between two passes a gate - the last
 *   workgroup of the pass' last kernel - announces the finished pass' sums (stored by the kernels in pinned memory) through
 *   a sequence word, polls a second word until the calling thread has published the next pass' control block - state,
 *   search / reuse, or stop; written straight into device memory when the BAR allows it - and copies it to where the pass
 *   kernels read it. The n x n algebra of esekfom.hpp:521-720 stays on the calling thread
 *   (half of it runs while the GPU is busy with the pass); what disappears is the synchronise / launch round trip per pass.
 * MALIO_UPDATE_HOST: one pass at a time - launch, synchronise, algebra, launch (what a pass hook, per-pass profiling,
 *   the node-sharded update and the M < n rows path use). Bit-identical to GATED.
 * MALIO_UPDATE_DEVICE: the algebra too on the GPU (one workgroup after every pass, which also decides what the next pass
 *   is): ONE chain the host waits for once. The serial 35-step eliminations are latency-bound on a GPU - measured 3x the
 *   time of the other two - so this is the mode for a host that must not be in the loop, not the fast one; differs from
 *   the others by the device's libm (sin/cos/atan, <= 2 ulp), i.e. ~1e-9 in the state at 1e5 points. */
enum { MALIO_UPDATE_DEVICE = 0, MALIO_UPDATE_HOST = 1, MALIO_UPDATE_GATED = 2 };
int malio_set_update_mode(malio_handle_t h, int mode);
/* The update without the host in the loop (what MALIO_UPDATE_DEVICE is for), split in two calls: `begin` enqueues every
 * pass of the loop and the n x n algebra of every iteration as kernels and returns at once; the calling thread is free
 * (~0.75 ms at BASELINE config 2: a gated update takes 0.16 ms but keeps the thread spinning) until `end` waits and
 * hands out state, covariance and stats exactly as malio_update_iterated does. Between the two calls nothing else may
 * be called on the handle. `end` may return MALIO_SMALL_M_FALLBACK (x, P untouched): call malio_update_iterated then. */
int malio_update_iterated_begin(malio_handle_t h, const malio_state_t *x, const double *P);
int malio_update_iterated_end(malio_handle_t h, malio_state_t *x, double *P, int *stats);

/* h_dyn_share is a plain function in the reference (esekfom.hpp:130,512): whatever it does - tracing, or finding the
 * map changed under it - happens once per pass on the calling thread. fn(pass, user) is called before every
 * measurement pass of malio_update_iterated[_node] with the 0-based pass number; NULL removes it. */
int malio_set_pass_hook(malio_handle_t h, void (*fn)(int pass, void *user), void *user);

/* The localization weight of laserMapping.cpp:745-756 from N^T N = sum c_i^2 n_i n_i^T (3 x 3 symmetric, given as
 * xx yy zz xy xz yz): w = sigma_3 / sigma_1 of h_x[:, 0:3] = sqrt(lambda_min / lambda_max), mapped onto
 * [localize_cov_min, localize_cov_max] between the two thresholds and clamped outside - what malio_measure applies
 * (as w^2) to the reduced normal equations. Pure host code (the device-resident update loop runs the same function). */
double malio_localize_weight(const double NtN6[6], double thresh_min, double thresh_max, double cov_min, double cov_max);

/* One iteration of the update loop AFTER its measurement pass (esekfom.hpp:521-720, the M >= n branch
 * :621-637), on the reduced normal equations. Pure host code, needs no handle and no GPU: a multi-GPU
 * driver calls it between its collectives. iter_index = loop index i of esekfom.hpp:509 (-1 ...
 * max_iteration-1); x: in = state the pass was evaluated at, out = x [+] dx; t_io = converged-iteration
 * counter (:658); converge_out = ekfom_data.converge for the next pass; done_out = 1 when P_out (n x n)
 * holds the posterior and the loop ends (:665-718); otherwise P_out receives the projected P_propagated of this
 * iteration (the value of the reference's member P_ after :531-572): a loop that runs out on invalid passes leaves
 * the filter with the one of its last valid iteration. limit: params.limit (0 = 0.001). */
int malio_ieskf_step(int lid_num, int max_iteration, double limit, int iter_index, malio_state_t *x,
                     const malio_state_t *x_propagated, const double *P_propagated, const double *HtRinvH,
                     const double *HtRinvh, int *t_io, int *converge_out, int *done_out, double *P_out);

/* ---- order of the scan inside the engine -------------------------------------------------------------------- */
/* The kernels want neighbouring queries to be neighbours in space. By default (AUTO) a scan handed over in host
 * buffers (malio_scan_set) is sorted by (LiDAR slot, map cell) at its first pass (~0.1 ms for 100 k points), while
 * a scan built by malio_scan_set_resident is used as it is: the voxel filter leaves every LiDAR's cloud sorted by
 * voxel index, which is coherent enough (the search pass is ~6 us slower, the sort is saved). SORT forces the sort
 * for both, KEEP skips it for malio_scan_set too - the caller then promises a spatially coherent cloud with the LiDAR
 * slots in ascending blocks (a cloud that is not grouped like that is sorted anyway). Results are identical sets of
 * rows either way; sums differ in the last bits with the order, as they do in the reference under OpenMP scheduling. */
enum { MALIO_SCAN_ORDER_AUTO = 0, MALIO_SCAN_ORDER_SORT = 1, MALIO_SCAN_ORDER_KEEP = 2 };
int malio_scan_order(malio_handle_t h, int mode);

/* ---- per-handle options ----------------------------------------------------------------------------------------- */
/* Everything that selects between two implementations of the same arithmetic (results are bit-identical either way) or
 * tunes a time-out lives on the handle and is set through this call - a ROS node configures its engine from its launch
 * file, not from the process environment. The reference has no counterpart (its knobs are compile-time: MP_PROC_NUM,
 * CMakeLists.txt:23-25). The environment variables of the same names (MALIO_FUSE, MALIO_MAINT_STREAM, ...) are read ONCE,
 * by malio_create, as the initial values: they exist for the A/B tools under tools/; a later malio_set_option wins.
 *   MALIO_OPT_FUSE             1 (default): search passes run as k_pass -> k_final_reduce, speculating on the previous pass'
 *                              extrema (laserMapping.cpp:625-628,646-647); 0: always three kernels.
 *   MALIO_OPT_SEARCH_SKIP      1: a search pass (ekfom_data.converge, laserMapping.cpp:582-591) that is not the first one of
 *                              its scan keeps the cached five neighbours of every point whose cache a distance certificate
 *                              proves unchanged (exact: same sets, same order, same bits as a full search; everything else
 *                              is searched). 0 (default): every search pass walks the lists for every point - the
 *                              certificate needs the 6th neighbour to be farther than the 5th by twice the point's motion,
 *                              and in a 0.5 m voxel map that gap is centimetres (1 % of the points kept
 *                              in a real update, +1 us per pass).
 *   MALIO_OPT_MAINT_STREAM     1 (default): list maintenance of map_add / map_incremental on a stream of its own.
 *   MALIO_OPT_MAPINC_SMALL     cap of map_incremental's one-read-back path in points (default 4096; 0: general path always).
 *   MALIO_OPT_GATE_PINNED      1: the gated update's control block goes through pinned host memory even under a large BAR.
 *   MALIO_OPT_GATE_TIMEOUT_MS  how long a gate waits for the calling thread before the update falls back to the
 *                              host-driven loop (default 200).
 *   MALIO_OPT_SCAN_SET_SYNC    1: malio_scan_set waits for the copy out of a page-locked cloud itself.
 *   MALIO_OPT_NL_FULL_BLOCKS   1: level-1 neighbour lists hold whole 3x3x3 blocks (takes effect at the next list build).
 *   MALIO_OPT_NL_SORTED        1 (default): level-1 neighbour lists are kept in order of distance from their cell's centre, so a
 *                              query's walk ends after the first 32 entries when those prove the rest irrelevant (exact: the
 *                              same five neighbours); 0: unordered lists, walked whole (rounds 2-4). Takes effect at the
 *                              next list build.
 *   MALIO_OPT_PROBE_CACHE      1 (default): a search pass remembers every point's level-1 directory probe (cell, list start and
 *                              length); the next search pass of the same scan - the lists unchanged - reuses it for every point
 *                              that is still in its cell (exact: the same list) instead of probing the directory again.
 *   MALIO_OPT_MAP_CELL_ORDER   1 (default): a (re)build of the search structures also puts the map array into cell order (columns of
 *                              level-1 cells, in the order the scan's grouping visits them), so that the five neighbours a query gathers for its plane fit share one
 *                              or two 128-byte lines whatever order the map was handed over or grew in; points added since
 *                              sit behind, in arrival order, until the next rebuild. Same neighbours either way (exact ties
 *                              of two map points' distances - (d2, slot) order - may resolve differently); malio_map_get and
 *                              the keeper rule's ties keep speaking of insertion order. Takes effect at the next rebuild.
 *   MALIO_OPT_EARLY_MIN_QUERIES  a walk of an ordered level-1 list ends early only in scans of at least this many queries
 *                              (default 32 768: below that the GPU has no queue of list lines to shorten and an unsettled
 *                              query's second trip is all there is; a tile shard counts the points it serves). 0: always -
 *                              the setting the edge-case tests run, and what a node of many small shards may want.
 *   MALIO_OPT_NODE_GATED       1 (default): malio_update_iterated_node / the node handle run the gated chain on every shard
 *                              (pass 0 through malio_measure_node, then one speculating pass per unit, the shards' rows
 *                              meeting in host memory between "sums seen" and "published"); 0: one pass at a time. Host
 *                              exchanges only - a MALIO_NODE_XCHG_RCCL node always runs one pass at a time. Must be the
 *                              same on every shard of a node.
 *   MALIO_OPT_DEBUG_*          test hooks: every guess of the extrema wrong / the host stalls before publishing pass 2.
 * Returns MALIO_ERR_BAD_ARG for an unknown option or a value outside its range. */
enum {
  MALIO_OPT_FUSE = 1,
  MALIO_OPT_SEARCH_SKIP = 2,
  MALIO_OPT_MAINT_STREAM = 3,
  MALIO_OPT_MAPINC_SMALL = 4,
  MALIO_OPT_GATE_PINNED = 5,
  MALIO_OPT_GATE_TIMEOUT_MS = 6,
  MALIO_OPT_SCAN_SET_SYNC = 7,
  MALIO_OPT_NL_FULL_BLOCKS = 8,
  MALIO_OPT_NODE_GATED = 9,
  MALIO_OPT_NL_SORTED = 10,
  MALIO_OPT_PROBE_CACHE = 11,
  MALIO_OPT_EARLY_MIN_QUERIES = 12,
  MALIO_OPT_MAP_CELL_ORDER = 13,
  MALIO_OPT_DEBUG_FUSE_BAD_GUESS = 100,
  MALIO_OPT_DEBUG_GATE_STALL_MS = 101,
  MALIO_OPT_DEBUG_NODE_GATED_RUNS = 102,  /* read-only (malio_get_option): updates of this shard through the gated chain ... */
  MALIO_OPT_DEBUG_NODE_GATED_REDONE = 103 /* ... and how many of them were handed back to the pass-by-pass loop */
};
int malio_set_option(malio_handle_t h, int option, double value);
int malio_get_option(malio_handle_t h, int option, double *value);
/* After a search pass: out4 = {points of the pass, points whose cached neighbours were kept (MALIO_OPT_SEARCH_SKIP),
 * points that walked the lists, 1 when the pass was allowed to skip at all}. */
int malio_debug_skip_stats(malio_handle_t h, int *out4);
/* The level-1 neighbour lists as the next search would find them: out4 = {lists, lists flagged as ordered (MALIO_OPT_NL_SORTED),
 * flagged lists that are NOT in order (always 0), live entries}. Builds stale lists first. lists - flagged = the lists a walk reads
 * whole: those above 256 entries (results are the same either way). A batch that touches more lists than its work list holds
 * (4 M items), or whose work list could not be allocated, has the whole directory put in order instead of its own lists. */
int malio_debug_list_order(malio_handle_t h, long long *out4);

/* ---- pinned host buffers (optional) ------------------------------------------------------------------------ */
/* Every entry point accepts ordinary (pageable) host memory, as the reference's std::vector / pcl clouds are. A copy
 * out of pageable memory is staged by the runtime in chunks and blocks the caller; out of page-locked memory it is one
 * DMA the call does not wait for. A caller that can choose where its clouds live (the raw points handed to
 * malio_undistort[_resident], the out_body of malio_scan_set_resident) gets them here; any device of the process. */
int malio_host_alloc(size_t bytes, void **out);
int malio_host_free(void *p);

/* ---- IMU propagation ------------------------------------------------------------- */
/* One kf.predict(dt, Q, in) (esekfom.hpp:388-492) with the process model of use-ikfom.hpp:67-112 (get_f, df_dx,
 * df_dw): x <- x (+) f(x, in) dt, P <- F P F^T + (dt f_w) Q (dt f_w)^T. predict_cont (:171-279) and back_predict
 * (:281-385) are the same step on kf's x_cont / x_unc and P_unc_: pass those instead. Pure host code, no handle and
 * no GPU. x, P (n x n row-major, n = 17+6 lid_num) in/out; P may be NULL to advance the state only. Q: 12 x 12
 * row-major process noise in the order ng, na, nbg, nba (use-ikfom.hpp:29-35, filled at IMU_Processing.hpp:325-330).
 * acc / gyro: input_ikfom.acc / .gyro (the caller applies the mean-acc scale of IMU_Processing.hpp:318). The loops
 * of IMU_Processing::UndistortPcl (:262-400) that call predict stay with the caller. */
int malio_predict(int lid_num, malio_state_t *x, double *P, double dt, const double *Q, const double *acc,
                  const double *gyro);

/* Chains of malio_predict steps on the device, up to four independent tracks side by side (one workgroup each, state and
 * covariance resident from the first step to the last): what ImuProcess::UndistortPcl runs one after the other per scan -
 * kf.predict on (x_, P_), predict_cont on (x_cont, P_unc_), back_predict on (x_unc, P_unc_)
 * (IMU_Processing.hpp:332,345,364,386,399) - is three tracks here. Track t starts from x[t] (in: start, out: end) and
 * P + t * n * n (in/out; P == NULL: states only) and takes K[t] steps whose dt / acc[3] / gyro[3] are concatenated in
 * track order; out_states (may be NULL) receives the state after every step in the same order (sum of K entries). Same
 * arithmetic as malio_predict (to the device's libm); lid_num is the handle's. */
int malio_predict_chain(malio_handle_t h, int n_tracks, malio_state_t *x, double *P, const int *K, const double *dt,
                        const double *acc, const double *gyro, const double *Q, malio_state_t *out_states);

/* ---- undistortion (IMU_Processing.hpp:475-507 + BsplineSE3.cpp:84-118) --------------------- */
/* Per-raw-point SE(3) cubic B-spline pose + rigid compensation into the LiDAR's own scan-end frame.
 * pts (in/out, sorted by curvature as :229-233): x,y,z rewritten, intensity <- uncertainty-interval
 * index; the first point is left untouched like the reference's loop bounds (:475-476).
 * knot_times[n_knots] / knot_poses[n_knots*16]: the spline's control poses (BsplineSE3.cpp:59-77),
 * absolute seconds. imu_stamps[n_imu]: imu_cov[k].first.first (ascending); cov_pointer0: value after :453-467.
 * out_entry_point[<= n_imu] (optional): indices of the points at which the reference pushes an uncertainty
 * entry (:484-494), in processing order (descending index); out_n_entries: how many. */
int malio_undistort(malio_handle_t h, malio_point_t *pts, int n, double lidar_beg_time, const double *knot_times,
                    const double *knot_poses, int n_knots, const double ext_q[4], const double ext_t[3],
                    const double end_q[4], const double end_t[3], const double *imu_stamps, int n_imu,
                    int cov_pointer0, int *out_entry_point, int *out_n_entries);
/* Host side of the trajectory spline (pure host, no handle):
 * ov_core::BsplineSE3::feed_trajectory (src/BsplineSE3.cpp:26-82): traj8[n][8] = t, p(3), q(x,y,z,w) ->
 * uniform 10 ms control poses (times + row-major 4x4). Returns MALIO_ERR_ALLOC when cap is too small. */
int malio_spline_feed(const double *traj8, int n, double *out_times, double *out_poses16, int cap, int *out_n);
/* ov_core::BsplineSE3::get_pose (src/BsplineSE3.cpp:84-118): 1 = success, 0 = the spline cannot bound
 * `timestamp` (p zeroed, like the reference). Needed on the host for the scan-end poses
 * (IMU_Processing.hpp:430,470) and the uncertainty-table entries (:488-492). */
int malio_spline_get_pose(const double *times, const double *poses16, int n, double timestamp, double q[4],
                          double p[3]);

/* ---- uncertainty tables on the host (a15; include/associate_uct.hpp) -------------------------------- */
/* compoundPoseWithCov(pose_1, cov_1, pose_2, cov_2, pose_cp, cov_cp, 2) (:85-142); the covariances travel inside
 * the poses; pose_cp may alias pose_2 (laserMapping.cpp:1043), with the reference's read/write order. */
int malio_compound_pose_cov(const malio_pose_t *pose_1, const malio_pose_t *pose_2, malio_pose_t *pose_cp);
/* compoundInvPoseWithCov(...) (:29-83): pose_cp = pose_1^-1 * pose_2 */
int malio_compound_inv_pose_cov(const malio_pose_t *pose_1, const malio_pose_t *pose_2, malio_pose_t *pose_cp);
/* evalPointUncertainty(pi, cov_point, pose) (:153-175): full 3x3 (row-major); the hot path only needs its trace. */
int malio_eval_point_uncertainty(const malio_point_t *pi, const malio_pose_t *pose, double cov_point[9]);

/* ---- multi-GPU staging: scan points sharded, map replicated -------------------- */
/* Stage 1: search/plane/gates + the local extrema into d_minmax, a device buffer of MALIO_MINMAX_LEN (8) doubles:
 *   [0..3] max_unit_cov, -min_unit_cov, max_R, -min_R  -> the caller all-reduces THESE FOUR with MAX
 *   [4]    local number of accepted points, [5] local search diagnostic (workgroups full of unmatched queries),
 *   [6..7] reserved                                     -> left as they are (per rank)
 * Stage 2: rows + local sums into d_sums (device, malio_sums_len() doubles: HtRinvH upper triangle, HtRinvh,
 * sum c^2 n n^T (6), M) -> all-reduce SUM. Finish: host-side weight/valid logic on the reduced sums (and the rank's
 * own copy of the 8 extrema words), fills `out` like malio_measure. */
#define MALIO_MINMAX_LEN 8
int malio_sums_len(malio_handle_t h);
int malio_measure_stage1(malio_handle_t h, const malio_state_t *s, int converge, double *d_minmax);
int malio_measure_stage2(malio_handle_t h, const double *d_minmax, double *d_sums);
/* Speculative variant (one exchange per pass): call malio_measure_stage1 with d_minmax = NULL (no fold launch), then
 * this with the extrema GUESSED from the previous pass in d_minmax_in; it weights the rows with the guess and also
 * writes this shard's own MALIO_MINMAX_LEN words to d_minmax_out. The caller exchanges [sums | own extrema], forms
 * the true extrema (MAX over ranks of the first four words) and, if they differ from the guess, repeats this call
 * (or malio_measure_stage2) with the true values. */
int malio_measure_stage2_emit(malio_handle_t h, const double *d_minmax_in, double *d_minmax_out, double *d_sums);
int malio_measure_finish(malio_handle_t h, const double *sums_host, const double *minmax_host,
                         malio_measure_out_t *out);

/* The handle's result buffer: page-locked host memory that the kernels can write (host pointer, its device alias,
 * length in doubles >= malio_sums_len() + MALIO_MINMAX_LEN; exists once a scan was set). Passing the device alias as
 * d_sums / d_minmax to the stage calls makes the results land in host memory without a copy kernel: synchronise the
 * stream and read *host. malio_measure uses it itself, so do not mix the two styles between a stage 1 and its finish. */
int malio_result_buffer(malio_handle_t h, double **host, double **dev, int *len_doubles);

/* Exchange of the staged results between the ranks of ONE node through POSIX shared memory: what travels per pass is
 * [malio_sums_len() sums | MALIO_MINMAX_LEN extrema words] = 2.4 KB that the host needs (malio_measure_finish and
 * malio_ieskf_step run there), so a GPU collective would only add a device round trip to a latency-bound message.
 * name: "/..." (shm_open); the rank with create != 0 makes and zeroes the segment BEFORE the others are told the name
 * (e.g. by the launcher's rendezvous) and unlinks it on destroy. all_gather: every rank passes its row (row_doubles
 * doubles) and receives all rows in rank order - summing them in that order gives every rank the same bits.
 * timeout_s > 0 turns a missing rank into an error instead of a hang. Host code, no GPU involved. */
typedef struct malio_xchg *malio_xchg_t;
int malio_xchg_create(const char *name, int rank, int world, int row_doubles, int create, malio_xchg_t *out);
int malio_xchg_all_gather(malio_xchg_t x, const double *in, double *out_all, double timeout_s);
/* One exchange of the speculating pass in a single call: gathers every rank's row = [ns sums | extrema words ...],
 * forms the true extrema (MAX over ranks of words ns .. ns+3) into extrema4_out and, unless they differ bitwise from
 * guess4 (the extrema the rows were weighted with; NULL = do not check), the rank-ordered sum of the ns sums into
 * sums_out (may alias row_in). Returns MALIO_OK, 1 when the guess missed (sums_out untouched: weight the rows again
 * with extrema4_out and call this with guess4 = NULL), or an error. */
int malio_xchg_reduce(malio_xchg_t x, const double *row_in, int ns, const double *guess4, double *sums_out,
                      double *extrema4_out, double timeout_s);
int malio_xchg_row(malio_xchg_t x); /* row_doubles the exchange was created with */
/* The same exchange between the THREADS of one process (one per GPU: what the node handle below runs on): creates all
 * `world` endpoints at once over a private block; endpoint r is used by thread r only. */
int malio_xchg_create_local(int world, int row_doubles, malio_xchg_t *out_world);
/* Diagnostics: mean latency [us] of one malio_xchg_reduce between `world` native threads of this process over a local
 * exchange (no GPU work): what MALIO_NODE_XCHG_HOST adds to a pass. The slowest thread's figure. */
int malio_debug_xchg_latency(int world, int row_doubles, int iters, double *us_out);
/* ... and over RCCL (xGMI between the GPUs of a node, the network beyond): the producing kernels leave the row in HBM
 * (malio_xchg_device_row), ncclAllGather runs on the handle's stream right behind them, one copy brings all rows to
 * pinned memory, ONE stream synchronisation per exchange; the rows are then added in rank order on the host like
 * everywhere else, so every rank holds the same bits. unique_id128: MALIO_RCCL_ID_BYTES from malio_rccl_unique_id()
 * on one rank, handed to the others by whatever launched them (torch.distributed / MPI / a file). Collective. */
#define MALIO_RCCL_ID_BYTES 128
int malio_rccl_unique_id(void *out128);
int malio_xchg_create_rccl(const void *unique_id128, int rank, int world, int row_doubles, int device, malio_xchg_t *out);
int malio_xchg_device_row(malio_xchg_t x, double **d_row);
/* 0 = shared memory (processes), 1 = local (threads), 2 = RCCL */
int malio_xchg_kind(malio_xchg_t x);
/* malio_xchg_reduce for an RCCL exchange whose device row was filled by work queued on `stream`; own_words_out (may
 * be NULL) receives this rank's own words after the four extrema. */
int malio_xchg_reduce_stream(malio_xchg_t x, void *stream, int ns, const double *guess4, double *sums_out,
                             double *extrema4_out, double *own_words_out);
/* Creator only, once every rank has opened the segment: removes the name (the mappings stay), so that nothing is left
 * in /dev/shm however the job ends. */
int malio_xchg_unlink(malio_xchg_t x);
int malio_xchg_destroy(malio_xchg_t x);

/* One pass over a scan sharded across the ranks of one node, in one call: malio_measure with scan-global extrema and
 * normal equations summed over all ranks (every rank gets the same bits and runs malio_ieskf_step on them). x must
 * have been created with row_doubles = malio_sums_len(h) + MALIO_MINMAX_LEN. The first pass after a scan was set
 * exchanges the extrema, then the sums; later passes weight their rows with the previous pass' extrema and need ONE
 * exchange (two again when the extrema moved). stats2 (may be NULL): passes that needed one / two exchanges so far. */
int malio_measure_node(malio_handle_t h, malio_xchg_t x, const malio_state_t *s, int converge, malio_measure_out_t *out,
                       int *stats2);

int malio_node_stats(malio_handle_t h, int *stats2); /* the two counters of malio_measure_node */
/* malio_update_iterated with malio_measure_node as h_dyn_share: every rank calls it with the same x and P and gets
 * the same posterior (bit for bit). Returns MALIO_SMALL_M_FALLBACK when fewer points than states were accepted
 * (esekfom.hpp:574-582 needs the rows of every rank: not a sharded path).
 * With a host-memory exchange (malio_xchg_create / _create_local) and MALIO_OPT_NODE_GATED on (the default) the passes after
 * the first run as the gated chain of malio_update_iterated - enqueued one ahead, a gate between two passes - and the ranks'
 * rows meet between a pass' sums and the next pass' control block; same results, bit for bit, as one pass at a time (what
 * an RCCL exchange, a pass hook, per-pass profiling or MALIO_OPT_FUSE = 0 select). The option, the update mode and
 * MALIO_OPT_FUSE must be the same on every rank: the ranks decide from them, each for itself, which loop to run. */
int malio_update_iterated_node(malio_handle_t h, malio_xchg_t x, malio_state_t *state, double *P, double R, int *stats,
                               double *solve_time);

/* ---- map sharded by space (BASELINE config 4) ------------------------------------------------ */
/* Makes this handle shard `rank` of `world`: space is cut into cubic tiles of edge tile_m (0 = 16 m), a tile belongs to
 * the shard its hashed coordinates name. Call before malio_map_build. From then on
 *   - malio_map_build / malio_map_add are handed the WHOLE map / all new points on every shard and keep the points of
 *     the shard's own tiles plus a 2.3 m halo (> sqrt(5) m, the radius beyond which laserMapping.cpp:587 rejects), by
 *     whole down-sampling voxels; malio_map_delete_boxes deletes within the shard;
 *   - malio_scan_set is handed the WHOLE scan on every shard; a SEARCH pass serves the points whose world point falls
 *     into an own tile (every shard computes the same bits, hence the same owner), REUSE passes keep serving those;
 *   - sums and extrema cover the served points: exchange them between the shards exactly as for a sharded scan
 *     (malio_measure_node / the stage calls). No neighbour merge is needed: the halo makes every acceptable 5-NN local,
 *     and results equal those of one handle holding the whole map (tests/test_partition.py).
 * malio_scan_get then only holds values for the served points: malio_scan_owned tells which (1 = served here). */
int malio_set_partition(malio_handle_t h, int rank, int world, float tile_m);
/* ... with the tiles' shape: MALIO_TILE_CUBES (what malio_set_partition sets) or MALIO_TILE_COLUMNS - a tile is the whole
 * vertical column over its tile_m x tile_m square: it has no neighbour above or below, so a shard's halo is the 2.3 m rim
 * around its squares only. On a ground vehicle's map (a few tens of metres high, hundreds wide) 8 shards then store 1.8 x the
 * map instead of 2.6 x (profiles/round5/r05_tile_shards.txt); ownership, exactness and the balance are as for cubes. */
enum { MALIO_TILE_CUBES = 0, MALIO_TILE_COLUMNS = 1 };
int malio_set_partition_shape(malio_handle_t h, int rank, int world, float tile_m, int shape);
int malio_scan_owned(malio_handle_t h, uint8_t *owned);

/* ---- several GPUs behind ONE handle, called from ONE thread ("multi-GPU handled inside") ------------- */
/* The reference drives the whole path from its single main thread (laserMapping.cpp:985-1060); malio_node_* is the
 * same interface as the malio_* calls above for a caller that owns n_gpus GPUs of one node: one worker thread per GPU
 * inside the library, the caller posts one call at a time. devices: HIP ordinals (NULL = 0 .. n_gpus-1; an ordinal may
 * repeat - several shards on one GPU - except with RCCL). partition:
 *   MALIO_PART_SCAN   map replicated on every GPU, the scan cut into n_gpus contiguous shards
 *   MALIO_PART_TILES  map sharded by spatial tiles of edge tile_m (0 = 16 m) with a halo, every GPU is handed the whole
 *                     scan and serves the points of its own tiles (malio_set_partition; BASELINE config 4)
 *   MALIO_PART_COLUMNS  the same with column-shaped tiles (malio_set_partition_shape, MALIO_TILE_COLUMNS): no halo above and
 *                     below a tile - the shape to use for a map that is much wider than it is high
 * exchange: how the per-pass [sums | extrema] rows (2.4 KB per GPU) meet - MALIO_NODE_XCHG_HOST: through host memory (the
 * rows are consumed by the host: the n x n filter algebra runs on the caller's thread), MALIO_NODE_XCHG_RCCL:
 * ncclAllGather over xGMI on the GPUs' streams. Either way the rows are added in GPU order: results do not depend on
 * timing, and equal those of one GPU given the whole scan and map up to the order of the final additions. */
typedef struct malio_node *malio_node_t;
enum { MALIO_PART_SCAN = 0, MALIO_PART_TILES = 1, MALIO_PART_COLUMNS = 2 };
enum { MALIO_NODE_XCHG_HOST = 0, MALIO_NODE_XCHG_RCCL = 1 };
int malio_node_create(const malio_params_t *params, int n_gpus, const int *devices, int partition, int exchange,
                      float tile_m, malio_node_t *out);
int malio_node_destroy(malio_node_t nd);
const char *malio_node_last_error(malio_node_t nd);
int malio_node_gpus(malio_node_t nd);
int malio_node_handle(malio_node_t nd, int rank, malio_handle_t *out); /* the per-GPU handle (diagnostics, profiling) */
/* == malio_map_build / malio_map_add / malio_map_delete_boxes on every GPU (a tile shard keeps its part);
 * out arrays (optional) take one value per GPU */
int malio_node_map_build(malio_node_t nd, const malio_point_t *pts, int n);
int malio_node_map_size(malio_node_t nd, int *out_sizes);
int malio_node_map_add(malio_node_t nd, const malio_point_t *pts, int n, int downsample_on, int *out_added);
int malio_node_map_delete_boxes(malio_node_t nd, const malio_box_t *boxes, int nb, int *out_deleted);
/* ikdtree.flatten / ikdtree.size for the whole node: a replica answers for all; tile shards contribute the points of their
 * own tiles (halo copies are not counted twice), rank after rank. */
int malio_node_map_get(malio_node_t nd, malio_point_t *out, int cap, int *out_n);
int malio_node_map_total(malio_node_t nd, int *out_size);
/* malio_voxel_downsample on the node (GPU 0) */
int malio_node_voxel_downsample(malio_node_t nd, const malio_point_t *pts, int n, float leaf, int normal_mode,
                                malio_point_t *out, int cap, int *out_n);
/* == malio_scan_set / malio_measure (no rows path) / malio_update_iterated / malio_scan_get / malio_set_pass_hook */
int malio_node_scan_set(malio_node_t nd, const malio_point_t *feats_down_body, int n, const malio_pose_t *const *pose_unc,
                        const int *pose_unc_len, const malio_pose_t *temporal_comp);
int malio_node_measure(malio_node_t nd, const malio_state_t *s, int converge, malio_measure_out_t *out);
int malio_node_update_iterated(malio_node_t nd, malio_state_t *x, double *P, double R, int *stats, double *solve_time);
int malio_node_scan_get(malio_node_t nd, float *normal_y, malio_point_t *nearest, int *nearest_count, uint8_t *selected,
                        float *res_last, float *world_xyz, float *normvec4);
/* map_incremental() on the node (see malio_map_incremental): every GPU classifies the scan points it serves, the two
 * lists are merged in scan order and handed to every GPU (a replica takes all, a tile shard what it stores). No
 * Nearest_Points cross PCIe. out_counts3: |PointToAdd|, |PointNoNeedDownsample|, GPU 0's return value of the first
 * Add_Points (the reference's value when the map is replicated). */
int malio_node_map_incremental(malio_node_t nd, const malio_state_t *state_point, int flg_EKF_inited,
                               const float *world_normal_y, int *out_counts3);
/* The resident front end on the node: LiDAR `lid` is undistorted and voxel-filtered on GPU lid % n_gpus (the L clouds of a
 * scan side by side instead of one after the other), the filtered clouds are concatenated in LiDAR order on the host and
 * installed on every GPU like malio_node_scan_set. Arguments as malio_undistort_resident / malio_scan_set_resident.
 * malio_node_scan_set_resident CONSUMES the resident clouds of the scan whether it succeeds or not: after an error from any
 * GPU (the filtered parts of the others are gone with the call) the caller undistorts the scan's clouds again. */
int malio_node_undistort_resident(malio_node_t nd, int lid, const malio_point_t *pts, int n, double lidar_beg_time,
                                  const double *knot_times, const double *knot_poses, int n_knots, const double ext_q[4],
                                  const double ext_t[3], const double end_q[4], const double end_t[3],
                                  const double *imu_stamps, int n_imu, int cov_pointer0, int *out_entry_point,
                                  int *out_n_entries, malio_point_t *out_entry_pts);
int malio_node_scan_set_resident(malio_node_t nd, float leaf, int normal_mode, const malio_pose_t *const *pose_unc,
                                 const int *pose_unc_len, const malio_pose_t *temporal_comp, malio_point_t *out_body, int cap,
                                 int *out_n);
/* malio_nearest_search on the node: replicas share the queries; a tile shard answers the queries of its tiles (it stores
 * every map point within 2.3 m of them; refused when 2 * cell_size is larger). */
int malio_node_nearest_search(malio_node_t nd, const malio_point_t *queries, int n, int k, malio_point_t *out_pts,
                              float *out_d2, int *out_count);
int malio_node_set_pass_hook(malio_node_t nd, void (*fn)(int pass, void *user), void *user);
int malio_node_set_option(malio_node_t nd, int option, double value); /* malio_set_option on every GPU's handle */
int malio_node_exchange_stats(malio_node_t nd, int *stats2); /* passes that needed one / two exchanges so far */
/* Summed over the node's shards: out4 = [updates that ran the gated chain, how many of those were handed back to the
 * pass-by-pass loop, gate time-outs (MALIO_OPT_GATE_TIMEOUT_MS), 0]. A node whose second number keeps growing has shards
 * whose gates starve (several shards on one device, a descheduled worker thread): MALIO_OPT_NODE_GATED = 0 is cheaper. */
int malio_node_update_stats(malio_node_t nd, int *out4);
/* shard geometry, host code (no GPU): which shard serves each of n world points (xyz [n][3]) / whether shard `rank`
 * stores each of n map points */
int malio_part_owner(const float *xyz, int n, int world, float tile_m, int *out_owner);
int malio_part_stores(const float *xyz, int n, int rank, int world, float tile_m, float filter_size_map, uint8_t *out_stores);
/* ... for either tile shape (MALIO_TILE_CUBES: the two above) */
int malio_part_owner_shape(const float *xyz, int n, int world, float tile_m, int shape, int *out_owner);
int malio_part_stores_shape(const float *xyz, int n, int rank, int world, float tile_m, int shape, float filter_size_map,
                            uint8_t *out_stores);

/* ---- measurement ------------------------------------------------------------------------------ */
/* Names/durations [ms] of the kernels of the last malio_measure / stage call, from hipEvents recorded
 * on the handle's stream. names: up to cap pointers to static strings. Returns count via *out_n. */
int malio_last_kernel_times(malio_handle_t h, const char **names, float *ms, int cap, int *out_n);
/* Enable/disable per-kernel event timing (off by default: events add launch latency). */
int malio_set_profiling(malio_handle_t h, int on);

/* Diagnostics: out8 = {level-1 directory cells, map points at the last list build, level-2 directory cells,
 * full list rebuilds so far, map changes applied to the lists in place so far, deleted slots awaiting compaction,
 * tombstoned points in the lists, map slots in use}. */
int malio_debug_counters(malio_handle_t h, int *out8);
/* {passes that ran as ONE kernel (the extrema of laserMapping.cpp:625-628,646-647 guessed from the previous pass of the
 * scan), guesses that held, guesses that missed (the rows of that pass were redone with the true extrema), gated updates
 * redone by the host-driven loop after a gate timed out}. MALIO_FUSE=0 in the environment disables the one-kernel pass. */
int malio_debug_fuse_stats(malio_handle_t h, int *out4);
/* After a search pass: out8[k] = scan points with k map points inside the sqrt(5) m acceptance radius (k = 0..5),
 * [6] = 0, [7] = points another shard serves. */
int malio_debug_nfound_hist(malio_handle_t h, int *out8);

#ifdef __cplusplus
}
#endif
#endif /* MALIO_H_ */
