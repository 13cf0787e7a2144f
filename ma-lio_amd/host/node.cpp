// The node handle: several GPUs behind ONE handle, called from ONE thread - the shape of the reference's integration
// (h_share_model and the ikd-Tree calls all come from laserMapping.cpp's single main thread, :985-1060) and of
// SURVEY.md §8b ("multi-GPU handled inside").
//
// One worker thread per GPU owns that GPU's malio handle (a Ctx: map shard or replica, scan shard, stream); the caller's
// thread posts one job at a time and waits for all workers. A measurement pass is malio_measure_node on every worker:
// the workers exchange their [sums | extrema] rows among themselves - through a private block of memory (the 2.4 KB are
// consumed by the host: the n x n filter algebra runs once, on the caller's thread) or through RCCL - and every worker
// ends up with the same reduced normal equations, added in rank order. Two ways to split the work:
//   MALIO_PART_SCAN   map replicated, scan cut into contiguous shards (every BASELINE map fits one GPU many times over)
//   MALIO_PART_TILES  map sharded by spatial tiles with a halo, every worker sees the whole scan and serves the points
//                     of its own tiles (BASELINE config 4; malio_set_partition)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <array>
#include <thread>
#include <vector>
#include <sched.h>
#include "../csrc/malio_internal.hpp"

namespace {

struct Worker {
  int rank = 0, device = 0;
  malio_handle_t h = nullptr;
  malio_xchg_t x = nullptr;
  std::thread th;
  int rc = 0;
  malio_measure_out_t out;
  // scan shard (MALIO_PART_SCAN): points [lo, hi) of the caller's cloud
  int lo = 0, hi = 0;
};

}  // namespace

struct malio_node {
  malio_params_t prm{};
  int n = 0, partition = MALIO_PART_SCAN, exchange = MALIO_NODE_XCHG_HOST;
  int columns = 0;  // MALIO_PART_COLUMNS: partition == MALIO_PART_TILES with column-shaped tiles
  float tile_m = 0.f;
  std::vector<Worker> w;
  std::string err;
  int N = 0;  // points of the current scan
  // job hand-off: the caller bumps `seq` after storing `job`; a worker runs it when it sees a new value and bumps `done`
  std::function<int(Worker &)> job;
  std::atomic<uint64_t> seq{0};
  std::atomic<int> done{0};
  std::atomic<bool> quit{false};
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<int> sleepers{0};
  void (*pass_hook)(int, void *) = nullptr;
  void *pass_hook_user = nullptr;
  // resident front end: LiDAR l's raw cloud lives on GPU l % n until malio_node_scan_set_resident consumes it
  bool res_has[MALIO_MAX_LIDAR] = {false, false, false, false};
  int res_n[MALIO_MAX_LIDAR] = {0, 0, 0, 0};
  std::vector<malio_point_t> body;  // feats_down_body of the current resident scan (page-locked would not pay: one copy per scan)

  int run(const std::function<int(Worker &)> &j) {  // all workers, in parallel; first non-zero status wins (errors first)
    job = j;
    done.store(0, std::memory_order_relaxed);
    seq.fetch_add(1, std::memory_order_release);
    if (sleepers.load(std::memory_order_acquire) > 0) {
      std::lock_guard<std::mutex> lk(mu);
      cv.notify_all();
    }
    unsigned spins = 0;
    while (done.load(std::memory_order_acquire) < n) {
      if (++spins < 4096)
        __builtin_ia32_pause();
      else
        spins = 0, sched_yield();
    }
    int rc = 0;
    for (auto &k : w)
      if (k.rc < 0) return k.rc;
    for (auto &k : w)
      if (k.rc > rc) rc = k.rc;
    return rc;
  }

  void loop(Worker &me) {
    (void)hipSetDevice(me.device);
    uint64_t seen = 0;
    while (true) {
      // passes of one update follow each other within tens of microseconds: spin for a while, then sleep
      const auto t0 = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (seq.load(std::memory_order_acquire) == seen && !quit.load(std::memory_order_relaxed)) {
        if (++spins < 2048) {
          __builtin_ia32_pause();
          continue;
        }
        spins = 0;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
          std::unique_lock<std::mutex> lk(mu);
          sleepers.fetch_add(1, std::memory_order_acq_rel);
          cv.wait_for(lk, std::chrono::milliseconds(50),
                      [&] { return seq.load(std::memory_order_acquire) != seen || quit.load(std::memory_order_relaxed); });
          sleepers.fetch_sub(1, std::memory_order_acq_rel);
        } else {
          sched_yield();
        }
      }
      if (quit.load(std::memory_order_relaxed) && seq.load(std::memory_order_acquire) == seen) return;
      seen = seq.load(std::memory_order_acquire);
      me.rc = job(me);
      done.fetch_add(1, std::memory_order_release);
    }
  }
};

using malio::Ctx;

extern "C" {

int malio_node_create(const malio_params_t *params, int n_gpus, const int *devices, int partition, int exchange,
                      float tile_m, malio_node_t *out) {
  if (!params || !out || n_gpus < 1 || n_gpus > 64) return MALIO_ERR_BAD_ARG;
  if (partition != MALIO_PART_SCAN && partition != MALIO_PART_TILES && partition != MALIO_PART_COLUMNS) return MALIO_ERR_BAD_ARG;
  const int columns = partition == MALIO_PART_COLUMNS ? 1 : 0;  // (from here on "tiles" of either shape: nd->columns says which)
  if (columns) partition = MALIO_PART_TILES;
  if (exchange != MALIO_NODE_XCHG_HOST && exchange != MALIO_NODE_XCHG_RCCL) return MALIO_ERR_BAD_ARG;
  *out = nullptr;
  // The HIP runtime is initialised HERE, on the calling thread, before any worker exists: when this call is the first
  // HIP use of the process, n workers entering hipGetDeviceCount at once race inside the runtime's one-time
  // initialisation and some of them are told there is no device.
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MALIO_ERR_NO_DEVICE;
  for (int r = 0; r < n_gpus; r++) {
    const int d = devices ? devices[r] : r;
    if (d < 0 || d >= ndev) return MALIO_ERR_NO_DEVICE;
  }
  malio_node *nd = new malio_node();
  nd->prm = *params, nd->n = n_gpus, nd->partition = partition, nd->columns = columns, nd->exchange = exchange, nd->tile_m = tile_m;
  nd->w.resize(n_gpus);
  const int row = params->lid_num * 97 + MALIO_MINMAX_LEN;
  std::vector<malio_xchg_t> xs(n_gpus, nullptr);
  if (exchange == MALIO_NODE_XCHG_HOST && malio_xchg_create_local(n_gpus, row, xs.data()) != MALIO_OK) {
    delete nd;
    return MALIO_ERR_ALLOC;
  }
  char uid[MALIO_RCCL_ID_BYTES];
  if (exchange == MALIO_NODE_XCHG_RCCL && malio_rccl_unique_id(uid) != MALIO_OK) {
    delete nd;
    return MALIO_ERR_HIP;
  }
  for (int r = 0; r < n_gpus; r++) {
    Worker &k = nd->w[r];
    k.rank = r, k.device = devices ? devices[r] : r, k.x = xs[r];
    k.th = std::thread([nd, r] { nd->loop(nd->w[r]); });
  }
  // handles (and RCCL communicators: ncclCommInitRank is collective) are created by the threads that will use them
  const void *uidp = uid;
  // Shards that share a device share its hardware queues (4 per process by default; every shard owns two streams): the gate
  // of one shard's unit, polling at the head of a queue, can hold up another shard's unit behind it until it times out -
  // correct (the time-out hands the update to the pass-by-pass loop, MALIO_OPT_GATE_TIMEOUT_MS) but slow. Two or three
  // shards per device - the test configurations - were measured to get by (profiles/round4/r04j_node_gated.txt) and keep
  // the gated chain; from four on the node updates pass by pass unless MALIO_OPT_NODE_GATED is set again. Production is
  // one shard per GPU. A node that keeps falling back shows in malio_node_update_stats.
  int most = 0;
  for (int r = 0; r < n_gpus; r++) {
    int same = 0;
    for (int q = 0; q < n_gpus; q++) same += nd->w[q].device == nd->w[r].device;
    most = same > most ? same : most;
  }
  const bool gated_ok = most <= 3;
  int rc = nd->run([nd, row, uidp, gated_ok](Worker &k) -> int {
    int rc = malio_create(&nd->prm, k.device, &k.h);
    if (rc != MALIO_OK) return rc;
    if (!gated_ok && (rc = malio_set_option(k.h, MALIO_OPT_NODE_GATED, 0.0)) != MALIO_OK) return rc;
    if (nd->partition == MALIO_PART_TILES &&
        (rc = malio_set_partition_shape(k.h, k.rank, nd->n, nd->tile_m, nd->columns ? MALIO_TILE_COLUMNS : MALIO_TILE_CUBES)) != MALIO_OK)
      return rc;
    if (nd->exchange == MALIO_NODE_XCHG_RCCL) rc = malio_xchg_create_rccl(uidp, k.rank, nd->n, row, k.device, &k.x);
    return rc;
  });
  if (rc != MALIO_OK) {
    malio_node_destroy(nd);
    return rc;
  }
  *out = nd;
  return MALIO_OK;
}

int malio_node_destroy(malio_node_t nd) {
  if (!nd) return MALIO_ERR_BAD_ARG;
  nd->run([](Worker &k) -> int {
    if (k.x) malio_xchg_destroy(k.x);
    if (k.h) malio_destroy(k.h);
    k.x = nullptr, k.h = nullptr;
    return 0;
  });
  nd->quit.store(true);
  {
    std::lock_guard<std::mutex> lk(nd->mu);
    nd->cv.notify_all();
  }
  for (auto &k : nd->w)
    if (k.th.joinable()) k.th.join();
  delete nd;
  return MALIO_OK;
}

const char *malio_node_last_error(malio_node_t nd) {
  if (!nd) return "null node";
  for (auto &k : nd->w)
    if (k.rc < 0 && k.h) return malio_last_error(k.h);
  return nd->err.c_str();
}

int malio_node_gpus(malio_node_t nd) { return nd ? nd->n : 0; }

int malio_node_handle(malio_node_t nd, int rank, malio_handle_t *out) {
  if (!nd || !out || rank < 0 || rank >= nd->n) return MALIO_ERR_BAD_ARG;
  *out = nd->w[rank].h;
  return MALIO_OK;
}

int malio_node_set_pass_hook(malio_node_t nd, void (*fn)(int, void *), void *user) {
  if (!nd) return MALIO_ERR_BAD_ARG;
  nd->pass_hook = fn, nd->pass_hook_user = user;
  return MALIO_OK;
}

int malio_node_set_option(malio_node_t nd, int option, double value) {  // malio_set_option on every GPU's handle
  if (!nd) return MALIO_ERR_BAD_ARG;
  return nd->run([=](Worker &k) { return malio_set_option(k.h, option, value); });
}

// ---- map: every GPU is handed the whole call; a tile shard keeps its part (malio_set_partition) ----------------------
int malio_node_map_build(malio_node_t nd, const malio_point_t *pts, int n) {
  if (!nd || !pts || n <= 0) return MALIO_ERR_BAD_ARG;
  return nd->run([=](Worker &k) { return malio_map_build(k.h, pts, n); });
}

int malio_node_map_size(malio_node_t nd, int *out_sizes /*[n_gpus]*/) {
  if (!nd || !out_sizes) return MALIO_ERR_BAD_ARG;
  for (int r = 0; r < nd->n; r++)
    if (int rc = malio_map_size(nd->w[r].h, &out_sizes[r])) return rc;
  return MALIO_OK;
}

int malio_node_map_add(malio_node_t nd, const malio_point_t *pts, int n, int downsample_on, int *out_added /*[n_gpus]*/) {
  if (!nd || n < 0 || (n > 0 && !pts)) return MALIO_ERR_BAD_ARG;
  return nd->run([=](Worker &k) { return malio_map_add(k.h, pts, n, downsample_on, out_added ? &out_added[k.rank] : nullptr); });
}

int malio_node_map_delete_boxes(malio_node_t nd, const malio_box_t *boxes, int nb, int *out_deleted /*[n_gpus]*/) {
  if (!nd || nb < 0 || (nb > 0 && !boxes)) return MALIO_ERR_BAD_ARG;
  return nd->run([=](Worker &k) { return malio_map_delete_boxes(k.h, boxes, nb, out_deleted ? &out_deleted[k.rank] : nullptr); });
}

// ikdtree.flatten / ikdtree.size on the node: a replica answers for all; tile shards answer with the points of their OWN
// tiles (a point in a halo is stored by several shards and owned by one), rank after rank.
int malio_node_map_get(malio_node_t nd, malio_point_t *out, int cap, int *out_n) {
  if (!nd || !out_n || cap < 0 || (cap > 0 && !out)) return MALIO_ERR_BAD_ARG;
  if (nd->partition == MALIO_PART_SCAN) {
    int rc = MALIO_OK;
    nd->run([&](Worker &k) -> int {
      if (k.rank == 0) rc = malio_map_get(k.h, out, cap, out_n);
      return MALIO_OK;
    });
    return rc;
  }
  const int G = nd->n;
  std::vector<std::vector<malio_point_t>> part(G);
  int rc = nd->run([&](Worker &k) -> int {
    int n = 0;
    int r = malio_map_get(k.h, nullptr, 0, &n);
    if (r != MALIO_OK) return r;
    std::vector<malio_point_t> all((size_t)std::max(n, 1));
    if ((r = malio_map_get(k.h, all.data(), n, &n)) != MALIO_OK) return r;
    malio::PartView pv;
    pv.rank = k.rank, pv.world = G, pv.inv_tile = 1.0f / (nd->tile_m > 0.f ? nd->tile_m : (nd->columns ? 24.f : 16.f)), pv.columns = nd->columns;
    pv.lat_k = malio::part_lattice_k(G);
    for (int i = 0; i < n; i++)
      if (all[i].x < 1e8f && malio::part_owns(pv, all[i].x, all[i].y, all[i].z)) part[k.rank].push_back(all[i]);  // (1e9: an empty shard's placeholder)
    return MALIO_OK;
  });
  if (rc != MALIO_OK) return rc;
  size_t total = 0;
  for (auto &v : part) total += v.size();
  *out_n = (int)total;
  size_t w = 0;
  for (auto &v : part)
    for (auto &p : v)
      if (w < (size_t)cap) out[w++] = p;
  return MALIO_OK;
}
int malio_node_map_total(malio_node_t nd, int *out_size) {
  if (!nd || !out_size) return MALIO_ERR_BAD_ARG;
  if (nd->partition == MALIO_PART_SCAN) return malio_map_size(nd->w[0].h, out_size);
  return malio_node_map_get(nd, nullptr, 0, out_size);
}
// pcl::VoxelGrid::filter on the node: a stateless service, GPU 0 renders it
int malio_node_voxel_downsample(malio_node_t nd, const malio_point_t *pts, int n, float leaf, int normal_mode, malio_point_t *out,
                                int cap, int *out_n) {
  if (!nd) return MALIO_ERR_BAD_ARG;
  int rc = MALIO_OK;
  nd->run([&](Worker &k) -> int {
    if (k.rank == 0) rc = malio_voxel_downsample(k.h, pts, n, leaf, normal_mode, out, cap, out_n);
    return MALIO_OK;
  });
  return rc;
}

// ---- scan -----------------------------------------------------------------------------------------------------------
int malio_node_scan_set(malio_node_t nd, const malio_point_t *body, int n, const malio_pose_t *const *pose_unc,
                        const int *pose_unc_len, const malio_pose_t *temporal_comp) {
  if (!nd || !body || n <= 0 || !pose_unc || !pose_unc_len) return MALIO_ERR_BAD_ARG;
  if (nd->partition == MALIO_PART_SCAN && n < nd->n) return MALIO_ERR_BAD_ARG;
  nd->N = n;
  for (int r = 0; r < nd->n; r++) {
    Worker &k = nd->w[r];
    if (nd->partition == MALIO_PART_SCAN)
      k.lo = (int)((long long)n * r / nd->n), k.hi = (int)((long long)n * (r + 1) / nd->n);
    else
      k.lo = 0, k.hi = n;
  }
  return nd->run([=](Worker &k) { return malio_scan_set(k.h, body + k.lo, k.hi - k.lo, pose_unc, pose_unc_len, temporal_comp); });
}

// ---- resident front end on the node (malio_undistort_resident / malio_scan_set_resident, include/malio.h) ------------------
// The L raw clouds of a scan do not depend on each other until they are concatenated (IMU_Processing.hpp:475-507 per
// LiDAR, laserMapping.cpp:966-983): LiDAR l is undistorted and voxel-filtered on GPU l % n - L GPUs work side by side
// on what one GPU does one LiDAR after the other - and only the FILTERED clouds (a sixth of the raw points) cross PCIe:
// to the host once, concatenated in LiDAR order (:982), and from there to every GPU as malio_node_scan_set does.
int malio_node_undistort_resident(malio_node_t nd, int lid, const malio_point_t *pts, int n, double lidar_beg_time,
                                  const double *knot_times, const double *knot_poses, int n_knots, const double ext_q[4],
                                  const double ext_t[3], const double end_q[4], const double end_t[3],
                                  const double *imu_stamps, int n_imu, int cov_pointer0, int *out_entry_point,
                                  int *out_n_entries, malio_point_t *out_entry_pts) {
  if (!nd || lid < 0 || lid >= nd->prm.lid_num) return MALIO_ERR_BAD_ARG;
  const int owner = lid % nd->n;
  const int rc = nd->run([=](Worker &k) -> int {
    if (k.rank != owner) return MALIO_OK;
    return malio_undistort_resident(k.h, lid, pts, n, lidar_beg_time, knot_times, knot_poses, n_knots, ext_q, ext_t, end_q, end_t,
                                    imu_stamps, n_imu, cov_pointer0, out_entry_point, out_n_entries, out_entry_pts);
  });
  if (rc == MALIO_OK) nd->res_has[lid] = true, nd->res_n[lid] = n;
  return rc;
}

int malio_node_scan_set_resident(malio_node_t nd, float leaf, int normal_mode, const malio_pose_t *const *pose_unc,
                                 const int *pose_unc_len, const malio_pose_t *temporal_comp, malio_point_t *out_body, int cap,
                                 int *out_n) {
  if (!nd || !pose_unc || !pose_unc_len || !out_n || cap < 0 || (cap > 0 && !out_body)) return MALIO_ERR_BAD_ARG;
  const int L = nd->prm.lid_num, G = nd->n;
  // every GPU that holds raw clouds filters them (its handle's own resident call; the scan it installs on the way is
  // replaced below) and hands its part of feats_down_body to the host
  std::vector<std::vector<malio_point_t>> part(G);
  std::vector<int> pn(G, 0);
  int rc = nd->run([&](Worker &k) -> int {
    int mine = 0;
    for (int l = k.rank; l < L; l += G)
      if (nd->res_has[l]) mine += nd->res_n[l];
    if (mine == 0) return MALIO_OK;
    part[k.rank].resize((size_t)mine);
    return malio_scan_set_resident(k.h, leaf, normal_mode, pose_unc, pose_unc_len, temporal_comp, part[k.rank].data(), mine,
                                   &pn[k.rank]);
  });
  for (int l = 0; l < MALIO_MAX_LIDAR; l++) nd->res_has[l] = false, nd->res_n[l] = 0;  // consumed (or lost with the error)
  if (rc != MALIO_OK) return rc;
  // concatenate in LiDAR order (:982): a GPU's part holds its LiDARs ascending, each point carries its LiDAR in `intensity`
  size_t total = 0;
  for (int r = 0; r < G; r++) total += (size_t)pn[r];
  *out_n = (int)total;
  if (total == 0) return MALIO_ERR_NO_SCAN;
  nd->body.resize(total);
  std::vector<size_t> at(G, 0);
  size_t w = 0;
  for (int l = 0; l < L; l++) {
    const int r = l % G;
    size_t &a = at[r];
    const size_t a0 = a;
    while (a < (size_t)pn[r] && (int)part[r][a].intensity == l) a++;
    memcpy(nd->body.data() + w, part[r].data() + a0, sizeof(malio_point_t) * (a - a0));
    w += a - a0;
  }
  if (w != total) {
    nd->err = "malio_node_scan_set_resident: a filtered cloud does not carry its LiDAR number";
    return MALIO_ERR_BAD_ARG;
  }
  if (out_body && cap > 0) memcpy(out_body, nd->body.data(), sizeof(malio_point_t) * std::min(total, (size_t)cap));
  return malio_node_scan_set(nd, nd->body.data(), (int)total, pose_unc, pose_unc_len, temporal_comp);
}

// ikdtree.Nearest_Search, batched (malio_nearest_search): a replica answers a contiguous share of the queries; a tile
// shard the queries of its own tiles - it stores every map point within PART_HALO = 2.3 m of them, and the search radius
// is 2 * cell_size (2.25 m at the default edge; a larger edge is refused here).
int malio_node_nearest_search(malio_node_t nd, const malio_point_t *queries, int n, int k, malio_point_t *out_pts, float *out_d2,
                              int *out_count) {
  if (!nd || !queries || n <= 0 || k < 1 || k > 5 || !out_pts || !out_d2 || !out_count) return MALIO_ERR_BAD_ARG;
  const int G = nd->n;
  if (nd->partition == MALIO_PART_SCAN)
    return nd->run([=](Worker &w) -> int {
      const int lo = (int)((long long)n * w.rank / G), hi = (int)((long long)n * (w.rank + 1) / G);
      if (hi <= lo) return MALIO_OK;
      return malio_nearest_search(w.h, queries + lo, hi - lo, k, out_pts + (size_t)lo * k, out_d2 + (size_t)lo * k, out_count + lo);
    });
  const float cell = nd->prm.cell_size > 0.f ? nd->prm.cell_size : 1.125f;
  if (2.f * cell > malio::PART_HALO) {
    nd->err = "malio_node_nearest_search: search radius 2 * cell_size exceeds the halo of a tile shard";
    return MALIO_ERR_BAD_ARG;
  }
  malio::PartView pv;
  pv.rank = 0, pv.world = G, pv.inv_tile = 1.0f / (nd->tile_m > 0.f ? nd->tile_m : (nd->columns ? 24.f : 16.f)), pv.columns = nd->columns;
  pv.lat_k = malio::part_lattice_k(G);
  std::vector<std::vector<int>> idx(G);
  for (int i = 0; i < n; i++) idx[malio::part_owner_of(pv, queries[i].x, queries[i].y, queries[i].z)].push_back(i);
  return nd->run([&](Worker &w) -> int {
    const std::vector<int> &mine = idx[w.rank];
    const int m = (int)mine.size();
    if (m == 0) return MALIO_OK;
    std::vector<malio_point_t> q((size_t)m), o((size_t)m * k);
    std::vector<float> d2((size_t)m * k);
    std::vector<int> cnt((size_t)m);
    for (int j = 0; j < m; j++) q[j] = queries[mine[j]];
    const int rc = malio_nearest_search(w.h, q.data(), m, k, o.data(), d2.data(), cnt.data());
    if (rc != MALIO_OK) return rc;
    for (int j = 0; j < m; j++) {
      memcpy(out_pts + (size_t)mine[j] * k, o.data() + (size_t)j * k, sizeof(malio_point_t) * k);
      memcpy(out_d2 + (size_t)mine[j] * k, d2.data() + (size_t)j * k, sizeof(float) * k);
      out_count[mine[j]] = cnt[j];
    }
    return MALIO_OK;
  });
}

int malio_node_measure(malio_node_t nd, const malio_state_t *s, int converge, malio_measure_out_t *out) {
  if (!nd || !s || !out) return MALIO_ERR_BAD_ARG;
  if (out->h_x || out->h || out->R) {
    nd->err = "malio_node_measure: the rows path is a single-GPU path";
    return MALIO_ERR_BAD_ARG;
  }
  const int rc = nd->run([=](Worker &k) {
    memset(&k.out, 0, sizeof(k.out));
    return malio_measure_node(k.h, k.x, s, converge, &k.out, nullptr);
  });
  if (rc < 0) return rc;
  *out = nd->w[0].out;  // every worker holds the same reduced result, bit for bit
  return rc;
}

int malio_node_update_iterated(malio_node_t nd, malio_state_t *x, double *P, double R, int *stats, double *solve_time) {
  if (!nd || !x || !P) return MALIO_ERR_BAD_ARG;
  if (solve_time) *solve_time = 0;
  if (nd->pass_hook) {
    // h_dyn_share is a plain function in the reference: the hook runs on the CALLING thread before every pass, so the
    // loop of esekfom.hpp:509 stays here and every pass is a job of its own for the workers
    malio::PassFn pass = [nd](const malio_state_t *s, int converge, malio_measure_out_t *mo) -> int {
      return malio_node_measure(nd, s, converge, mo);
    };
    return malio::ieskf_update_fn(nd->prm, pass, nullptr, nd->N, nd->pass_hook, nd->pass_hook_user, x, P, R, stats, solve_time);
  }
  // ONE job: every worker runs the whole loop on its own copy of (x, P) - pass, exchange, the n x n algebra - and the
  // workers meet in the exchange of every pass and nowhere else. The reduced sums are identical on every worker bit for
  // bit (rows added in rank order), hence so are the iterates and the posterior: worker 0's are handed out. Against a job
  // per pass this takes the caller's thread off every pass' critical path (two hand-offs and the wake-up of n workers
  // per pass, the algebra while every GPU idles).
  const int n = 17 + 6 * nd->prm.lid_num;
  std::vector<malio_state_t> xs((size_t)nd->n, *x);
  std::vector<std::vector<double>> Ps((size_t)nd->n, std::vector<double>(P, P + (size_t)n * n));
  std::vector<std::array<int, 4>> st((size_t)nd->n, std::array<int, 4>{0, 0, 0, 0});
  std::vector<double> solve((size_t)nd->n, 0.0);
  const int rc = nd->run([&](Worker &k) -> int {
    return malio_update_iterated_node(k.h, k.x, &xs[k.rank], Ps[k.rank].data(), R, st[k.rank].data(), &solve[k.rank]);
  });
  if (rc != MALIO_OK) return rc;  // (MALIO_SMALL_M_FALLBACK included: x and P untouched, as on one engine)
  for (int r = 1; r < nd->n; r++)
    if (memcmp(&xs[r], &xs[0], sizeof(malio_state_t)) != 0 || memcmp(Ps[r].data(), Ps[0].data(), sizeof(double) * (size_t)n * n) != 0) {
      nd->err = "malio_node_update_iterated: the workers' iterates differ (the exchange must hand every rank the same rows)";
      return MALIO_ERR_HIP;
    }
  *x = xs[0];
  memcpy(P, Ps[0].data(), sizeof(double) * (size_t)n * n);
  if (stats) memcpy(stats, st[0].data(), sizeof(int) * 4);
  if (solve_time) *solve_time = solve[0];
  return MALIO_OK;
}

int malio_node_exchange_stats(malio_node_t nd, int *stats2) {
  if (!nd || !stats2) return MALIO_ERR_BAD_ARG;
  return malio_node_stats(nd->w[0].h, stats2);
}

int malio_node_update_stats(malio_node_t nd, int *out4) {
  if (!nd || !out4) return MALIO_ERR_BAD_ARG;
  out4[0] = out4[1] = out4[2] = out4[3] = 0;
  for (int r = 0; r < nd->n; r++) {
    double runs = 0, redone = 0;
    int fs[4] = {0, 0, 0, 0};
    int rc = malio_get_option(nd->w[r].h, MALIO_OPT_DEBUG_NODE_GATED_RUNS, &runs);
    if (rc == MALIO_OK) rc = malio_get_option(nd->w[r].h, MALIO_OPT_DEBUG_NODE_GATED_REDONE, &redone);
    if (rc == MALIO_OK) rc = malio_debug_fuse_stats(nd->w[r].h, fs);
    if (rc != MALIO_OK) return rc;
    out4[0] += (int)runs, out4[1] += (int)redone, out4[2] += fs[3];
  }
  return MALIO_OK;
}

// Side effects in the caller's scan order, merged from the GPUs: a scan shard returns its own range, a tile shard the
// points it served (malio_scan_owned).
int malio_node_scan_get(malio_node_t nd, float *normal_y, malio_point_t *nearest, int *nearest_count, uint8_t *selected,
                        float *res_last, float *world_xyz, float *normvec4) {
  if (!nd) return MALIO_ERR_BAD_ARG;
  if (nd->N <= 0) return MALIO_ERR_NO_SCAN;
  if (nd->partition == MALIO_PART_SCAN)
    return nd->run([=](Worker &k) {
      const size_t o = (size_t)k.lo;
      return malio_scan_get(k.h, normal_y ? normal_y + o : nullptr, nearest ? nearest + 5 * o : nullptr,
                            nearest_count ? nearest_count + o : nullptr, selected ? selected + o : nullptr,
                            res_last ? res_last + o : nullptr, world_xyz ? world_xyz + 3 * o : nullptr,
                            normvec4 ? normvec4 + 4 * o : nullptr);
    });
  // tiles: every worker fills private arrays, then copies the entries of the points it served (disjoint between workers)
  const int N = nd->N;
  return nd->run([=](Worker &k) -> int {
    std::vector<float> ny(normal_y ? N : 0), rl(res_last ? N : 0), wx(world_xyz ? 3 * (size_t)N : 0), nv(normvec4 ? 4 * (size_t)N : 0);
    std::vector<malio_point_t> nr(nearest ? 5 * (size_t)N : 0);
    std::vector<int> nc(nearest_count ? N : 0);
    std::vector<uint8_t> sl(selected ? N : 0), own(N);
    int rc = malio_scan_get(k.h, normal_y ? ny.data() : nullptr, nearest ? nr.data() : nullptr, nearest_count ? nc.data() : nullptr,
                            selected ? sl.data() : nullptr, res_last ? rl.data() : nullptr, world_xyz ? wx.data() : nullptr,
                            normvec4 ? nv.data() : nullptr);
    if (rc != MALIO_OK) return rc;
    if ((rc = malio_scan_owned(k.h, own.data())) != MALIO_OK) return rc;
    for (int i = 0; i < N; i++) {
      if (!own[i]) continue;
      if (normal_y) normal_y[i] = ny[i];
      if (nearest) memcpy(nearest + 5 * (size_t)i, nr.data() + 5 * (size_t)i, sizeof(malio_point_t) * 5);
      if (nearest_count) nearest_count[i] = nc[i];
      if (selected) selected[i] = sl[i];
      if (res_last) res_last[i] = rl[i];
      if (world_xyz) memcpy(world_xyz + 3 * (size_t)i, wx.data() + 3 * (size_t)i, sizeof(float) * 3);
      if (normvec4) memcpy(normvec4 + 4 * (size_t)i, nv.data() + 4 * (size_t)i, sizeof(float) * 4);
    }
    return MALIO_OK;
  });
}

// map_incremental() (laserMapping.cpp:398-446) on the node: every GPU classifies the scan points it serves (its range of
// the scan, or the points of its tiles) from the neighbours of ITS last search pass - no Nearest_Points cross PCIe - and
// hands back its two lists; they are merged in scan order on the caller's thread (the order Add_Points' sequential
// semantics inside a voxel depends on) and every GPU is handed both merged lists: a replica takes all of them, a tile
// shard what it stores (own tiles + halo). counts3: |PointToAdd|, |PointNoNeedDownsample|, and the return value of the
// first Add_Points on GPU 0 (the reference's value for MALIO_PART_SCAN; for MALIO_PART_TILES GPU 0's share of it).
int malio_node_map_incremental(malio_node_t nd, const malio_state_t *state_point, int flg_EKF_inited,
                               const float *world_normal_y, int *out_counts3) {
  if (!nd || !state_point) return MALIO_ERR_BAD_ARG;
  if (nd->N <= 0) return MALIO_ERR_NO_SCAN;
  const int G = nd->n;
  struct Sel {
    std::vector<malio_point_t> pts;
    std::vector<int> idx;
    int cnt[2] = {0, 0};
  };
  std::vector<Sel> sel(G);
  int rc = nd->run([&](Worker &k) -> int {
    Sel &s = sel[k.rank];
    const int n = k.hi - k.lo;
    s.pts.resize((size_t)n), s.idx.resize((size_t)n);
    const int r = malio_map_incremental_select(k.h, state_point, flg_EKF_inited, world_normal_y ? world_normal_y + k.lo : nullptr,
                                               s.pts.data(), s.idx.data(), n, s.cnt);
    if (r != MALIO_OK) return r;
    for (int j = 0; j < s.cnt[0] + s.cnt[1]; j++) s.idx[j] += k.lo;  // positions in the caller's cloud
    return MALIO_OK;
  });
  if (rc != MALIO_OK) return rc;
  // merge by scan index: PointToAdd of all GPUs, then PointNoNeedDownsample (each GPU's lists are ascending already)
  std::vector<malio_point_t> lists[2];
  for (int part = 0; part < 2; part++) {
    std::vector<int> at(G);
    size_t total = 0;
    for (int r = 0; r < G; r++) at[r] = part == 0 ? 0 : sel[r].cnt[0], total += (size_t)sel[r].cnt[part];
    lists[part].reserve(total);
    while (lists[part].size() < total) {
      int best = -1, bidx = 0x7FFFFFFF;
      for (int r = 0; r < G; r++) {
        const int end = part == 0 ? sel[r].cnt[0] : sel[r].cnt[0] + sel[r].cnt[1];
        if (at[r] < end && sel[r].idx[at[r]] < bidx) best = r, bidx = sel[r].idx[at[r]];
      }
      lists[part].push_back(sel[best].pts[at[best]++]);
    }
  }
  std::vector<int> added(G, 0);
  const bool ds = (float)nd->prm.filter_size_map > 0.f;
  rc = nd->run([&](Worker &k) -> int {  // ikdtree.Add_Points(PointToAdd, true); ikdtree.Add_Points(PointNoNeedDownsample, false)
    int r = malio_map_add(k.h, lists[0].data(), (int)lists[0].size(), ds ? 1 : 0, &added[k.rank]);
    if (r != MALIO_OK) return r;
    return malio_map_add(k.h, lists[1].data(), (int)lists[1].size(), 0, nullptr);
  });
  if (out_counts3) out_counts3[0] = (int)lists[0].size(), out_counts3[1] = (int)lists[1].size(), out_counts3[2] = ds ? added[0] : 0;
  return rc;
}

// ---- shard geometry (host code: tests, and callers that want to know where a point lives) -------------------------
int malio_part_owner(const float *xyz, int n, int world, float tile_m, int *out_owner) {
  return malio_part_owner_shape(xyz, n, world, tile_m, MALIO_TILE_CUBES, out_owner);
}
int malio_part_owner_shape(const float *xyz, int n, int world, float tile_m, int shape, int *out_owner) {
  if (!xyz || !out_owner || n < 0 || world < 1 || (shape != MALIO_TILE_CUBES && shape != MALIO_TILE_COLUMNS)) return MALIO_ERR_BAD_ARG;
  malio::PartView p;
  p.rank = 0, p.world = world, p.inv_tile = 1.0f / (tile_m > 0.f ? tile_m : (shape == MALIO_TILE_COLUMNS ? 24.f : 16.f)), p.columns = shape == MALIO_TILE_COLUMNS;
  p.lat_k = malio::part_lattice_k(world);
  for (int i = 0; i < n; i++) out_owner[i] = (int)malio::part_owner_of(p, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  return MALIO_OK;
}
int malio_part_stores(const float *xyz, int n, int rank, int world, float tile_m, float filter_size_map, uint8_t *out_stores) {
  return malio_part_stores_shape(xyz, n, rank, world, tile_m, MALIO_TILE_CUBES, filter_size_map, out_stores);
}
int malio_part_stores_shape(const float *xyz, int n, int rank, int world, float tile_m, int shape, float filter_size_map,
                            uint8_t *out_stores) {
  if (!xyz || !out_stores || n < 0 || world < 1 || rank < 0 || rank >= world || (shape != MALIO_TILE_CUBES && shape != MALIO_TILE_COLUMNS))
    return MALIO_ERR_BAD_ARG;
  malio::PartView p;
  p.rank = rank, p.world = world, p.inv_tile = 1.0f / (tile_m > 0.f ? tile_m : (shape == MALIO_TILE_COLUMNS ? 24.f : 16.f)), p.columns = shape == MALIO_TILE_COLUMNS;
  p.lat_k = malio::part_lattice_k(world);
  for (int i = 0; i < n; i++) out_stores[i] = malio::part_stores(p, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], filter_size_map) ? 1 : 0;
  return MALIO_OK;
}

}  // extern "C"
