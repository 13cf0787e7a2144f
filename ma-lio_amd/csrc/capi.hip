// C ABI of libmalio_hip.so (include/malio.h). Thin: argument checks, host<->HBM staging, kernel
// orchestration on the handle's stream. No CPU fallback: without a gfx950 device malio_create fails.
#include <algorithm>
#include <cmath>
#include "malio_internal.hpp"
#include "build_id.h"  // MALIO_BUILD_ID (written by the Makefile)
#include "../host/manifold.hpp"

using namespace malio;

namespace malio {

__global__ void k_noop() {}
// malio_map_build on a tile shard: which points of the whole map does this shard store (keep[n] = 0 closes the scan)?
__global__ void __launch_bounds__(BLK) k_part_flags(const float4 *__restrict__ pts, int n, PartView part, float fs, u32 *keep) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i > n) return;
  keep[i] = (i < n && part_stores(part, pts[i].x, pts[i].y, pts[i].z, fs)) ? 1u : 0u;
}
__global__ void __launch_bounds__(BLK) k_part_compact(const float4 *__restrict__ src, const u32 *__restrict__ keep,
                                                      const u32 *__restrict__ pos, int n, float4 *dst) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i < n && keep[i]) dst[pos[i]] = src[i];
}

void prof_begin(Ctx *c) {
  c->ev_used = 0;
  c->ev_names.clear();
  if (!c->profiling) return;
  if (c->ev.empty()) {
    c->ev.resize(16);
    for (auto &e : c->ev) (void)hipEventCreate(&e);
  }
  // one empty kernel first: it absorbs the idle-queue dispatch latency, so that the first interval starts where
  // the first real kernel can start (intervals still carry ~2-3 us of marker overhead each compared with
  // rocprofv3's dispatch durations; bench.py reports them as they are - the pessimistic side)
  hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, c->stream);
  (void)hipEventRecord(c->ev[0], c->stream);
  c->ev_used = 1;
}
void prof_mark(Ctx *c, const char *name) {
  if (!c->profiling || c->ev_used == 0 || c->ev_used >= (int)c->ev.size()) return;
  (void)hipEventRecord(c->ev[c->ev_used], c->stream);
  c->ev_names.push_back(name);
  c->ev_used++;
}
void prof_end(Ctx *c) {
  if (!c->profiling || c->ev_used < 2) return;
  (void)hipEventSynchronize(c->ev[c->ev_used - 1]);
  c->last_ms.clear();
  c->last_names.clear();
  for (int k = 1; k < c->ev_used; k++) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, c->ev[k - 1], c->ev[k]);
    c->last_ms.push_back(ms);
    c->last_names.push_back(c->ev_names[k - 1]);
  }
}

// Fold one pose_unc entry so that trace(Sigma_p) is a quadratic form in p' = T (0.05 p, 1):
// G = [I | -[p']x | R],  Sigma = blkdiag(1e4 cov, 0.1 I)   (associate_uct.hpp:153-175)
static void fold_entry(const malio_pose_t &p, UncEntry &e) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) e.T[i * 4 + j] = p.T[i * 4 + j];
  double S[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) S[i][j] = p.cov[i * 6 + j] * 10000;
  double rr = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) rr += p.T[i * 4 + j] * p.T[i * 4 + j];
  e.k0 = S[0][0] + S[1][1] + S[2][2] + 0.1 * rr;
  auto vee = [](double M[3][3], double v[3]) {
    v[0] = M[1][2] - M[2][1], v[1] = M[2][0] - M[0][2], v[2] = M[0][1] - M[1][0];
  };
  double Str[3][3], Srt[3][3], Srr[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Str[i][j] = S[i][3 + j], Srt[i][j] = S[3 + i][j], Srr[i][j] = S[3 + i][3 + j];
  double a[3], b[3];
  vee(Str, a), vee(Srt, b);
  for (int k = 0; k < 3; k++) e.lin[k] = a[k] - b[k];
  double tr = Srr[0][0] + Srr[1][1] + Srr[2][2];
  e.Q[0] = tr - Srr[0][0], e.Q[1] = tr - Srr[1][1], e.Q[2] = tr - Srr[2][2];
  e.Q[3] = -0.5 * (Srr[0][1] + Srr[1][0]), e.Q[4] = -0.5 * (Srr[0][2] + Srr[2][0]),
  e.Q[5] = -0.5 * (Srr[1][2] + Srr[2][1]);
}

// Nearest_Points[i] in the caller's order: the five of the search pass where it found five inside sqrt(5) m, else the
// unrestricted 5-NN (far_knn5) - ikdtree.Nearest_Search has no radius (ikd_Tree.cpp:426-461)
__global__ void __launch_bounds__(BLK) k_gather_side(int N, const u32 *__restrict__ perm, const u32 *__restrict__ nbr,
                                                     const u32 *__restrict__ far, const unsigned char *__restrict__ nfound,
                                                     const float4 *__restrict__ map_pts, float4 *out_near /*[N][5]*/,
                                                     int *out_cnt /*[N] caller's order*/) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= N) return;
  u32 o = perm[i];
  const u32 *src = nfound[i] >= 5 ? nbr : far;
  int cnt = 0;
  for (int k = 0; k < 5; k++) {
    u32 j = src[(size_t)k * N + i];  // original map index
    cnt += j != 0xFFFFFFFFu;
    out_near[(size_t)o * 5 + k] = (j != 0xFFFFFFFFu) ? map_pts[j] : make_float4(0, 0, 0, 0);
  }
  out_cnt[o] = cnt;
}

}  // namespace malio

static int check(malio_handle_t h) { return h ? MALIO_OK : MALIO_ERR_BAD_ARG; }
static int rc_dev_row(malio_xchg_t x, double **row) { return malio_xchg_device_row(x, row); }

// pinned staging buffer of the upload paths, grown on demand and kept (hipHostMalloc/hipHostFree cost ~0.7 ms each)
int malio::host_stage(malio::Ctx *c, size_t bytes, void **out) {
  if (c->stage_pending) {  // malio_scan_set returns with its upload still in flight
    MALIO_HIP(hipStreamSynchronize(c->stream));
    c->stage_pending = false;
  }
  if (bytes > c->cap_stage) {
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr, c->cap_stage = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    MALIO_HIP(hipHostMalloc(&c->h_stage, want, hipHostMallocDefault));
    c->cap_stage = want;
  }
  *out = c->h_stage;
  return MALIO_OK;
}

extern "C" {

const char *malio_version(void) { return "malio-hip 0.1 (gfx950, ABI 1)"; }
const char *malio_build_id(void) { return MALIO_BUILD_ID; }

int malio_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

const char *malio_last_error(malio_handle_t h) { return h ? h->err.c_str() : "null handle"; }

// Initial option values: the environment variables of the A/B tools, read once per handle, here (include/malio.h,
// malio_set_option).
static void options_from_env(malio_handle_t c) {
  auto num = [](const char *name, double *v) {
    const char *e = getenv(name);
    if (!e || !*e) return false;
    *v = atof(e);
    return true;
  };
  static const struct { const char *env; int opt; } tab[] = {
      {"MALIO_FUSE", MALIO_OPT_FUSE}, {"MALIO_SEARCH_SKIP", MALIO_OPT_SEARCH_SKIP}, {"MALIO_MAINT_STREAM", MALIO_OPT_MAINT_STREAM},
      {"MALIO_MAPINC_SMALL", MALIO_OPT_MAPINC_SMALL}, {"MALIO_GATE_PINNED", MALIO_OPT_GATE_PINNED},
      {"MALIO_GATE_TIMEOUT_MS", MALIO_OPT_GATE_TIMEOUT_MS}, {"MALIO_SCAN_SET_SYNC", MALIO_OPT_SCAN_SET_SYNC},
      {"MALIO_NL_FULL_BLOCKS", MALIO_OPT_NL_FULL_BLOCKS}, {"MALIO_NL_SORTED", MALIO_OPT_NL_SORTED}, {"MALIO_PROBE_CACHE", MALIO_OPT_PROBE_CACHE}, {"MALIO_EARLY_MIN_QUERIES", MALIO_OPT_EARLY_MIN_QUERIES}, {"MALIO_MAP_CELL_ORDER", MALIO_OPT_MAP_CELL_ORDER}, {"MALIO_NODE_GATED", MALIO_OPT_NODE_GATED}, {"MALIO_DEBUG_FUSE_BAD_GUESS", MALIO_OPT_DEBUG_FUSE_BAD_GUESS},
      {"MALIO_DEBUG_GATE_STALL_MS", MALIO_OPT_DEBUG_GATE_STALL_MS}};
  for (const auto &t : tab) {
    double v;
    if (num(t.env, &v)) (void)malio_set_option(c, t.opt, v);  // (a value out of range is ignored)
  }
}

int malio_set_option(malio_handle_t h, int option, double value) {
  if (check(h) || !(value == value)) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  const bool is01 = value == 0.0 || value == 1.0;
  switch (option) {
    case MALIO_OPT_FUSE:
      if (!is01) return MALIO_ERR_BAD_ARG;
      c->fuse_enabled = (int)value;
      return MALIO_OK;
    case MALIO_OPT_SEARCH_SKIP:
      if (!is01) return MALIO_ERR_BAD_ARG;
      c->opt_search_skip = (int)value;
      return MALIO_OK;
    case MALIO_OPT_MAINT_STREAM:
      if (!is01) return MALIO_ERR_BAD_ARG;
      if ((int)value != c->maint_enabled) {  // whatever is queued on the maintenance stream first
        if (int rc = maint_join(c)) return rc;
        c->maint_enabled = (int)value;
      }
      return MALIO_OK;
    case MALIO_OPT_MAPINC_SMALL:
      if (value < 0.0 || value > 1e9) return MALIO_ERR_BAD_ARG;
      c->mapinc_small = (int)value;  // (clamped to the kernel's capacity where it is used)
      return MALIO_OK;
    case MALIO_OPT_GATE_PINNED:
      if (!is01) return MALIO_ERR_BAD_ARG;
      c->opt_gate_pinned = (int)value;  // takes effect at the next update (ensure_gate_buffers)
      return MALIO_OK;
    case MALIO_OPT_GATE_TIMEOUT_MS:
      if (value < 0.0 || value > 1e7) return MALIO_ERR_BAD_ARG;
      c->gate_timeout_ticks = (long long)(value * 1e5);  // 100 MHz; 0: the default (GATE_TIMEOUT_US)
      return MALIO_OK;
    case MALIO_OPT_SCAN_SET_SYNC:
      if (!is01) return MALIO_ERR_BAD_ARG;
      c->scan_set_sync = (int)value;
      return MALIO_OK;
    case MALIO_OPT_NL_FULL_BLOCKS:
      if (!is01) return MALIO_ERR_BAD_ARG;
      c->opt_nl_full_blocks = (int)value;  // takes effect at the next list build
      return MALIO_OK;
    case MALIO_OPT_NL_SORTED:
      if (!is01) return MALIO_ERR_BAD_ARG;
      c->opt_nl_sorted = (int)value;  // takes effect at the next list build
      return MALIO_OK;
    case MALIO_OPT_PROBE_CACHE:
      if (!is01) return MALIO_ERR_BAD_ARG;
      c->opt_probe_cache = (int)value;
      return MALIO_OK;
    case MALIO_OPT_NODE_GATED:
      if (!is01) return MALIO_ERR_BAD_ARG;
      c->opt_node_gated = (int)value;
      return MALIO_OK;
    case MALIO_OPT_MAP_CELL_ORDER:
      if (!is01) return MALIO_ERR_BAD_ARG;
      c->opt_map_cell_order = (int)value;  // takes effect at the next rebuild
      return MALIO_OK;
    case MALIO_OPT_EARLY_MIN_QUERIES:
      if (value < 0.0 || value > 2e9) return MALIO_ERR_BAD_ARG;
      c->opt_early_min_queries = (int)value;  // (read when a search pass is queued)
      return MALIO_OK;
    case MALIO_OPT_DEBUG_FUSE_BAD_GUESS:
      if (!is01) return MALIO_ERR_BAD_ARG;
      c->fuse_debug_bad_guess = value != 0.0;
      return MALIO_OK;
    case MALIO_OPT_DEBUG_GATE_STALL_MS:
      if (value < 0.0 || value > 1e4) return MALIO_ERR_BAD_ARG;
      c->gate_debug_stall_ms = (int)value;
      return MALIO_OK;
    default:
      c->err = "malio_set_option: unknown option";
      return MALIO_ERR_BAD_ARG;
  }
}

int malio_get_option(malio_handle_t h, int option, double *value) {
  if (check(h) || !value) return MALIO_ERR_BAD_ARG;
  const Ctx *c = h;
  switch (option) {
    case MALIO_OPT_FUSE: *value = c->fuse_enabled; return MALIO_OK;
    case MALIO_OPT_SEARCH_SKIP: *value = c->opt_search_skip; return MALIO_OK;
    case MALIO_OPT_MAINT_STREAM: *value = c->maint_enabled; return MALIO_OK;
    case MALIO_OPT_MAPINC_SMALL: *value = c->mapinc_small; return MALIO_OK;
    case MALIO_OPT_GATE_PINNED: *value = c->opt_gate_pinned; return MALIO_OK;
    case MALIO_OPT_GATE_TIMEOUT_MS: *value = (double)c->gate_timeout_ticks * 1e-5; return MALIO_OK;
    case MALIO_OPT_SCAN_SET_SYNC: *value = c->scan_set_sync; return MALIO_OK;
    case MALIO_OPT_NL_FULL_BLOCKS: *value = c->opt_nl_full_blocks; return MALIO_OK;
    case MALIO_OPT_NL_SORTED: *value = c->opt_nl_sorted; return MALIO_OK;
    case MALIO_OPT_PROBE_CACHE: *value = c->opt_probe_cache; return MALIO_OK;
    case MALIO_OPT_NODE_GATED: *value = c->opt_node_gated; return MALIO_OK;
    case MALIO_OPT_EARLY_MIN_QUERIES: *value = c->opt_early_min_queries; return MALIO_OK;
    case MALIO_OPT_MAP_CELL_ORDER: *value = c->opt_map_cell_order; return MALIO_OK;
    case MALIO_OPT_DEBUG_FUSE_BAD_GUESS: *value = c->fuse_debug_bad_guess ? 1.0 : 0.0; return MALIO_OK;
    case MALIO_OPT_DEBUG_GATE_STALL_MS: *value = c->gate_debug_stall_ms; return MALIO_OK;
    case MALIO_OPT_DEBUG_NODE_GATED_RUNS: *value = c->node_gated_runs; return MALIO_OK;  // (read-only counters)
    case MALIO_OPT_DEBUG_NODE_GATED_REDONE: *value = c->node_gated_redone; return MALIO_OK;
    default: return MALIO_ERR_BAD_ARG;
  }
}

int malio_create(const malio_params_t *params, int device, malio_handle_t *out) {
  if (!params || !out) return MALIO_ERR_BAD_ARG;
  if (params->lid_num < 1 || params->lid_num > MALIO_MAX_LIDAR) return MALIO_ERR_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return MALIO_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return MALIO_ERR_NO_DEVICE;
  malio_ctx *c = new malio_ctx();
  c->prm = *params;
  c->device = device;
  c->cell = params->cell_size > 0.f ? params->cell_size : 1.125f;
  if (c->cell < 0.25f) c->cell = 0.25f;  // level 2 is max(2 * cell, 2.25 m): it alone must cover the sqrt(5) m radius
  c->inv_cell = 1.0f / c->cell;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return MALIO_ERR_HIP;
  }
  c->stream = c->own_stream;
  options_from_env(c);
  *out = c;
  return MALIO_OK;
}

int malio_destroy(malio_handle_t h) {
  if (!h) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->maint_stream) (void)hipStreamSynchronize(c->maint_stream);  // (queued list maintenance, a staged copy: done before
  if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);    //  anything they touch is freed)
  auto fr = [](void *p) {
    if (p) (void)hipFree(p);
  };
  free_nlist(c->nl1);
  free_nlist(c->nl2);
  free_nl_scratch(c->nl_scratch);
  free_grid(c->gnew);
  free_dev_loop(c);
  c->arena.release_all();
  for (auto &rc : c->res) fr(rc.d);
  fr(c->d_map_in), fr(c->d_map_ord), fr(c->d_world4), fr(c->d_mmslots), fr(c->d_ny), fr(c->d_cert), fr(c->d_kept), fr(c->d_pcache);
  fr(c->d_map_alt), fr(c->d_raw), fr(c->d_packinfo), fr(c->d_sort_cnt), fr(c->d_del);
  if (c->h_packinfo) (void)hipHostFree(c->h_packinfo);
  fr(c->d_upload), fr(c->d_scan), fr(c->d_perm), fr(c->d_unc), fr(c->d_nbr), fr(c->d_plane), fr(c->d_pd2);
  fr(c->d_world), fr(c->d_ucov), fr(c->d_trace), fr(c->d_sel), fr(c->d_nfound), fr(c->d_partials);
  fr(c->d_sums), fr(c->d_rows), fr(c->d_tiles);
  if (c->h_sums) (void)hipHostFree(c->h_sums);
  if (c->h_res) (void)hipHostFree(c->h_res);
  if (c->h_node_mm) (void)hipHostFree(c->h_node_mm);
  if (c->d_node_mm) (void)hipFree(c->d_node_mm);
  if (c->h_mbox) (void)hipHostFree(c->h_mbox);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->h_minmax) (void)hipHostFree(c->h_minmax);
  for (auto &e : c->ev) (void)hipEventDestroy(e);
  if (c->ev_upload) (void)hipEventDestroy(c->ev_upload);
  maint_destroy(c);
  if (c->d_small_table) (void)hipFree(c->d_small_table);
  if (c->d_small_orig) (void)hipFree(c->d_small_orig);
  if (c->copy_stream) {
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipEventDestroy(c->ev_ahead), (void)hipEventDestroy(c->ev_ahead_free);
    (void)hipStreamDestroy(c->copy_stream);
  }
  if (c->d_ahead) (void)hipFree(c->d_ahead);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete h;
  return MALIO_OK;
}

int malio_set_stream(malio_handle_t h, void *hip_stream, int external) {
  if (check(h)) return MALIO_ERR_BAD_ARG;
  (void)hipStreamSynchronize(h->stream);
  h->stream = external ? (hipStream_t)hip_stream : h->own_stream;
  return MALIO_OK;
}

int malio_set_partition(malio_handle_t h, int rank, int world, float tile_m) {
  return malio_set_partition_shape(h, rank, world, tile_m, MALIO_TILE_CUBES);
}
int malio_set_partition_shape(malio_handle_t h, int rank, int world, float tile_m, int shape) {
  if (check(h) || world < 1 || rank < 0 || rank >= world || (shape != MALIO_TILE_CUBES && shape != MALIO_TILE_COLUMNS)) return MALIO_ERR_BAD_ARG;
  if (h->map_n > 0) {
    h->err = "malio_set_partition: call before malio_map_build";
    return MALIO_ERR_BAD_ARG;
  }
  if (!(tile_m > 0.f)) tile_m = shape == MALIO_TILE_COLUMNS ? 24.f : 16.f;  // (columns: 1.48 x the map on 8 shards, balance 1.05)
  if (tile_m < 4.f * (PART_HALO + 0.5f * (float)h->prm.filter_size_map)) return MALIO_ERR_BAD_ARG;  // part_touches: <= 8 tiles
  h->part.rank = rank, h->part.world = world, h->part.inv_tile = 1.0f / tile_m, h->part.columns = shape == MALIO_TILE_COLUMNS ? 1 : 0;
  h->part.lat_k = part_lattice_k(world);
  return MALIO_OK;
}

int malio_scan_owned(malio_handle_t h, uint8_t *owned) {
  if (check(h) || !owned) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  if (c->N <= 0 || !c->scan_sorted) return MALIO_ERR_NO_SCAN;
  MALIO_HIP(hipSetDevice(c->device));
  std::vector<u32> perm(c->N);
  std::vector<unsigned char> nf(c->N);
  MALIO_HIP(hipMemcpyAsync(perm.data(), c->d_perm, sizeof(u32) * c->N, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipMemcpyAsync(nf.data(), c->d_nfound, c->N, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  for (int i = 0; i < c->N; i++) owned[perm[i]] = nf[i] != 0xFD;  // NF_NOTMINE
  return MALIO_OK;
}

int malio_set_profiling(malio_handle_t h, int on) {
  if (check(h)) return MALIO_ERR_BAD_ARG;
  h->profiling = on != 0;
  return MALIO_OK;
}

int malio_last_kernel_times(malio_handle_t h, const char **names, float *ms, int cap, int *out_n) {
  if (check(h) || !out_n) return MALIO_ERR_BAD_ARG;
  int n = std::min<int>(cap, (int)h->last_ms.size());
  for (int i = 0; i < n; i++) {
    if (names) names[i] = h->last_names[i];
    if (ms) ms[i] = h->last_ms[i];
  }
  *out_n = n;
  return MALIO_OK;
}

// ---- map ------------------------------------------------------------------------------------------
int malio_map_build(malio_handle_t h, const malio_point_t *pts, int n) {
  if (check(h) || !pts || n <= 0) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  MALIO_HIP(hipSetDevice(c->device));
  if (int rcj = maint_join(c)) return rcj;  // (queued maintenance of the map this call replaces)
  (void)map_apply_finish(c);
  float4 *stage = nullptr;
  if (int rcs = host_stage(c, sizeof(float4) * (size_t)n + 16, (void **)&stage)) return rcs;
  for (int i = 0; i < n; i++) stage[i] = make_float4(pts[i].x, pts[i].y, pts[i].z, pts[i].normal_y);
  if (c->part.world > 1) {
    // this shard's part of the map: own tiles + halo, in the caller's order (ties between equal distances are broken by
    // map index, so the relative order must be the unsharded one). Every shard of a node is handed the whole map
    // (8 M points at BASELINE config 4): the ownership test runs on the GPU - flags, scan, stable compaction - instead
    // of 8 M part_stores() calls on this thread (same float arithmetic on host and device: malio_part_stores is the
    // host's view of it).
    ArenaScope sc(c->arena);
    float4 *d_all = nullptr;
    u32 *keep = nullptr, *kpos = nullptr, *tiles = nullptr;
    MALIO_HIP(sc.get(&d_all, (size_t)n));
    MALIO_HIP(sc.get(&keep, (size_t)n + 1));
    MALIO_HIP(sc.get(&kpos, (size_t)n + 1));
    MALIO_HIP(sc.get(&tiles, (size_t)(n + 1 + 1023) / 1024 + 2));
    MALIO_HIP(hipMemcpyAsync(d_all, stage, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_part_flags, dim3((n + 1 + BLK - 1) / BLK), dim3(BLK), 0, c->stream, d_all, n, c->part,
                       (float)c->prm.filter_size_map, keep);
    exclusive_scan_u32(c, keep, kpos, tiles, n + 1);
    u32 m = 0;
    MALIO_HIP(hipMemcpyAsync(&m, kpos + n, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    MALIO_HIP(hipStreamSynchronize(c->stream));
    c->part_sentinel = m == 0;
    const size_t need = m ? m : 1;
    if (need > c->cap_map_in) {
      if (c->d_map_in) (void)hipFree(c->d_map_in);
      c->d_map_in = nullptr;
      c->cap_map_in = need + need / 8 + 1024;
      MALIO_HIP(hipMalloc(&c->d_map_in, sizeof(float4) * c->cap_map_in));
    }
    if (m == 0) {  // an empty shard still answers "no neighbours" for its tiles
      stage[0] = make_float4(1e9f, 1e9f, 1e9f, 0.f);
      MALIO_HIP(hipMemcpyAsync(c->d_map_in, stage, sizeof(float4), hipMemcpyHostToDevice, c->stream));
    } else {
      hipLaunchKernelGGL(k_part_compact, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, c->stream, d_all, keep, kpos, n, c->d_map_in);
    }
    MALIO_HIP(hipStreamSynchronize(c->stream));  // (the arena's temporaries are handed back with this scope)
    c->map_n = (int)need;
    c->map_dead = 0;
    c->map_sorted_n = 0;  // (as uploaded: insertion order)
    c->map_epoch++;
    int rc = map_rebuild_search(c);
    (void)hipStreamSynchronize(c->stream);
    if (rc != MALIO_OK) c->map_n = 0;
    return rc;
  }
  if ((size_t)n > c->cap_map_in) {
    if (c->d_map_in) (void)hipFree(c->d_map_in);
    c->d_map_in = nullptr;
    c->cap_map_in = (size_t)n + (size_t)n / 8 + 1024;
    MALIO_HIP(hipMalloc(&c->d_map_in, sizeof(float4) * c->cap_map_in));
  }
  MALIO_HIP(hipMemcpyAsync(c->d_map_in, stage, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  c->map_n = n;
  c->map_dead = 0;
  c->map_sorted_n = 0;  // (as uploaded: insertion order)
  c->map_epoch++;
  int rc = map_rebuild_search(c);
  (void)hipStreamSynchronize(c->stream);
  if (rc != MALIO_OK) c->map_n = 0;
  return rc;
}

int malio_map_size(malio_handle_t h, int *out_size) {
  if (check(h) || !out_size) return MALIO_ERR_BAD_ARG;
  *out_size = h->map_n - h->map_dead - (h->part_sentinel ? 1 : 0);
  return MALIO_OK;
}

int malio_nearest_search(malio_handle_t h, const malio_point_t *queries, int n, int k, malio_point_t *out_pts,
                         float *out_d2, int *out_count) {
  if (check(h) || !queries || n <= 0 || k < 1 || k > 5) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  if (c->map_n - c->map_dead <= 0) return MALIO_ERR_NO_MAP;
  MALIO_HIP(hipSetDevice(c->device));
  std::vector<float4> hq(n);
  for (int i = 0; i < n; i++) hq[i] = make_float4(queries[i].x, queries[i].y, queries[i].z, 0.f);
  float4 *d_q = nullptr;
  u32 *d_idx = nullptr;
  float *d_d2 = nullptr;
  int *d_cnt = nullptr;
  MALIO_HIP(hipMalloc(&d_q, sizeof(float4) * (size_t)n));
  MALIO_HIP(hipMalloc(&d_idx, sizeof(u32) * (size_t)n * k));
  MALIO_HIP(hipMalloc(&d_d2, sizeof(float) * (size_t)n * k));
  MALIO_HIP(hipMalloc(&d_cnt, sizeof(int) * (size_t)n));
  MALIO_HIP(hipMemcpyAsync(d_q, hq.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  int rc = nearest_search(c, d_q, n, k, d_idx, d_d2, d_cnt);
  if (rc != MALIO_OK) return rc;
  std::vector<u32> idx((size_t)n * k);
  std::vector<float> d2((size_t)n * k);
  std::vector<int> cnt(n);
  MALIO_HIP(hipMemcpyAsync(idx.data(), d_idx, sizeof(u32) * idx.size(), hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipMemcpyAsync(d2.data(), d_d2, sizeof(float) * d2.size(), hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost, c->stream));
  std::vector<float4> mp(c->map_n);  // map array (slot index = map id)
  MALIO_HIP(hipMemcpyAsync(mp.data(), c->d_map_in, sizeof(float4) * mp.size(), hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; i++) {
    if (out_count) out_count[i] = cnt[i];
    for (int j = 0; j < k; j++) {
      size_t o = (size_t)i * k + j;
      if (out_d2) out_d2[o] = d2[o];
      if (out_pts) {
        malio_point_t p;
        memset(&p, 0, sizeof(p));
        if (idx[o] != 0xFFFFFFFFu) {
          float4 m = mp[idx[o]];
          p.x = m.x, p.y = m.y, p.z = m.z, p._pad0 = 1.f, p.normal_y = m.w;
        }
        out_pts[o] = p;
      }
    }
  }
  (void)hipFree(d_q), (void)hipFree(d_idx), (void)hipFree(d_d2), (void)hipFree(d_cnt);
  return MALIO_OK;
}

int malio_map_add(malio_handle_t h, const malio_point_t *pts, int n, int downsample_on, int *out_added) {
  if (check(h) || n < 0 || (n > 0 && !pts)) return MALIO_ERR_BAD_ARG;
  std::vector<float4> stage;
  stage.reserve((size_t)n);
  const float fs = (float)h->prm.filter_size_map;
  for (int i = 0; i < n; i++)  // a shard takes the points it stores (own tiles + halo, whole voxels): see part_stores
    if (part_stores(h->part, pts[i].x, pts[i].y, pts[i].z, fs)) stage.push_back(make_float4(pts[i].x, pts[i].y, pts[i].z, pts[i].normal_y));
  return map_add(h, stage.data(), (int)stage.size(), downsample_on, out_added);
}

int malio_map_delete_boxes(malio_handle_t h, const malio_box_t *boxes, int nb, int *out_deleted) {
  if (check(h) || nb < 0 || (nb > 0 && !boxes)) return MALIO_ERR_BAD_ARG;
  return map_delete_boxes(h, boxes, nb, out_deleted);
}

int malio_map_incremental(malio_handle_t h, const malio_state_t *state_point, int flg_EKF_inited,
                          const float *world_normal_y, int *out_counts3) {
  if (check(h) || !state_point) return MALIO_ERR_BAD_ARG;
  if (h->part.world > 1) {
    h->err = "malio_map_incremental: a map shard only knows the neighbours of its own points; the points to add must reach every shard (use the node handle, or malio_scan_get + malio_map_add on every shard)";
    return MALIO_ERR_BAD_ARG;
  }
  return map_incremental(h, state_point, flg_EKF_inited, world_normal_y, out_counts3);
}

int malio_map_incremental_select(malio_handle_t h, const malio_state_t *state_point, int flg_EKF_inited,
                                 const float *world_normal_y, malio_point_t *out_pts, int *out_index, int cap, int *out_counts2) {
  if (check(h) || !state_point || !out_counts2 || cap < 0 || (cap > 0 && !out_pts)) return MALIO_ERR_BAD_ARG;
  return map_incremental_select(h, state_point, flg_EKF_inited, world_normal_y, out_pts, out_index, cap, out_counts2);
}

int malio_map_get(malio_handle_t h, malio_point_t *out, int cap, int *out_n) {
  if (check(h) || !out_n || cap < 0 || (cap > 0 && !out)) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  *out_n = c->map_n - c->map_dead;
  if (cap <= 0 || c->map_n <= 0) return MALIO_OK;
  MALIO_HIP(hipSetDevice(c->device));
  if (int rcj = maint_join(c)) return rcj;
  std::vector<float4> mp((size_t)c->map_n);
  MALIO_HIP(hipMemcpyAsync(mp.data(), c->d_map_in, sizeof(float4) * mp.size(), hipMemcpyDeviceToHost, c->stream));
  // "map order" is INSERTION order, whatever order the slots are kept in (the array's first map_sorted_n slots are in cell order
  // since the last rebuild, d_map_ord holds their ranks): slot_of[rank] first, then the ranks in turn
  std::vector<u32> slot_of;
  if (c->map_sorted_n > 0) {
    std::vector<u32> ord((size_t)c->map_sorted_n);
    MALIO_HIP(hipMemcpyAsync(ord.data(), c->d_map_ord, sizeof(u32) * ord.size(), hipMemcpyDeviceToHost, c->stream));
    MALIO_HIP(hipStreamSynchronize(c->stream));
    slot_of.resize(ord.size());
    for (size_t i = 0; i < ord.size(); i++) slot_of[ord[i]] = (u32)i;
  }
  MALIO_HIP(hipStreamSynchronize(c->stream));
  int k = 0;
  for (size_t e = 0; e < mp.size() && k < cap; e++) {
    const size_t i = e < slot_of.size() ? slot_of[e] : e;
    if (std::isinf(mp[i].x)) continue;  // deleted slot
    malio_point_t p;
    memset(&p, 0, sizeof(p));
    p.x = mp[i].x, p.y = mp[i].y, p.z = mp[i].z, p._pad0 = 1.f, p.normal_y = mp[i].w;
    out[k++] = p;
  }
  return MALIO_OK;
}

int malio_decode_livox(malio_handle_t h, const unsigned char *records, int n_records, int n_scans, int point_filter_num,
                       double blind, int eof_point, malio_point_t *out, int cap, int *out_n, double *maximum_time) {
  if (check(h) || n_records < 0 || (n_records > 0 && !records) || point_filter_num < 1 || n_scans < 0 || !out_n || cap < 0 ||
      (cap > 0 && !out))
    return MALIO_ERR_BAD_ARG;
  return decode_livox(h, records, n_records, n_scans, point_filter_num, blind, eof_point, out, cap, out_n, maximum_time);
}

int malio_decode_ouster(malio_handle_t h, const unsigned char *records, int n_records, int point_filter_num, double blind,
                        float time_unit_scale, malio_point_t *out, int cap, int *out_n, double *maximum_time) {
  if (check(h) || n_records < 0 || (n_records > 0 && !records) || point_filter_num < 1 || !out_n || cap < 0 ||
      (cap > 0 && !out))
    return MALIO_ERR_BAD_ARG;
  return decode_ouster(h, records, n_records, point_filter_num, blind, time_unit_scale, out, cap, out_n, maximum_time);
}

int malio_decode_velodyne(malio_handle_t h, const unsigned char *data, int n_points, const malio_pc2_layout_t *layout,
                          int point_filter_num, double blind, float time_unit_scale, malio_point_t *out, int cap, int *out_n,
                          double *maximum_time) {
  if (check(h) || n_points < 0 || (n_points > 0 && !data) || !layout || point_filter_num < 1 || !out_n || cap < 0 ||
      (cap > 0 && !out))
    return MALIO_ERR_BAD_ARG;
  const malio_pc2_layout_t &l = *layout;
  const int need[3] = {l.off_x, l.off_y, l.off_z}, opt[2] = {l.off_intensity, l.off_time};
  if (l.point_step < 12) return MALIO_ERR_BAD_ARG;
  for (int k = 0; k < 3; k++)
    if (need[k] < 0 || need[k] + 4 > l.point_step) return MALIO_ERR_BAD_ARG;
  for (int k = 0; k < 2; k++)
    if (opt[k] >= 0 && opt[k] + 4 > l.point_step) return MALIO_ERR_BAD_ARG;
  if ((long long)n_points * l.point_step > 0x7FFFFFFFll * 4) return MALIO_ERR_BAD_ARG;
  return decode_velodyne(h, data, n_points, l, point_filter_num, blind, time_unit_scale, out, cap, out_n, maximum_time);
}

int malio_voxel_downsample(malio_handle_t h, const malio_point_t *pts, int n, float leaf, int normal_mode,
                           malio_point_t *out, int cap, int *out_n) {
  if (check(h) || !out_n || n < 0 || cap < 0 || (n > 0 && !pts) || (cap > 0 && !out)) return MALIO_ERR_BAD_ARG;
  if (normal_mode != MALIO_VOXEL_NORMAL_MEAN && normal_mode != MALIO_VOXEL_NORMAL_NORMALIZE) return MALIO_ERR_BAD_ARG;
  return voxel_downsample(h, pts, n, leaf, normal_mode, out, cap, out_n);
}

// ---- scan -----------------------------------------------------------------------------------------
// per-scan tables: folded pose_unc entries (trace as a quadratic form) and the temporal compensation
static int scan_tables(Ctx *c, const malio_pose_t *const *pose_unc, const int *pose_unc_len, const malio_pose_t *temporal_comp) {
  const int L = c->prm.lid_num;
  int tot = 0;
  for (int l = 0; l < L; l++) {
    if (pose_unc_len[l] < 2 || !pose_unc[l]) return MALIO_ERR_BAD_ARG;  // the reference indexes size()-2
    c->unc_off[l] = tot, c->unc_len[l] = pose_unc_len[l];
    tot += pose_unc_len[l];
  }
  // folded entries are staged in the pinned mailbox (words 4096..) when they fit, so that the copy is queued like
  // everything else instead of blocking the call (a few KB; the stream is synchronised by every pass, long before the
  // next scan overwrites the staging area)
  u32 *mb = nullptr;
  MALIO_HIP(mbox(c, &mb));
  const size_t bytes = sizeof(UncEntry) * (size_t)tot;
  const bool staged = bytes <= sizeof(u32) * (MBOX_WORDS - 4096);
  std::vector<UncEntry> ue_heap(staged ? 0 : tot);
  UncEntry *ue = staged ? reinterpret_cast<UncEntry *>(mb + 4096) : ue_heap.data();
  for (int l = 0; l < L; l++)
    for (int k = 0; k < pose_unc_len[l]; k++) fold_entry(pose_unc[l][k], ue[c->unc_off[l] + k]);
  if ((size_t)tot > c->cap_unc) {
    if (c->d_unc) (void)hipFree(c->d_unc);
    c->cap_unc = tot + 64;
    MALIO_HIP(hipMalloc(&c->d_unc, sizeof(UncEntry) * c->cap_unc));
  }
  if (staged)
    MALIO_HIP(hipMemcpyAsync(c->d_unc, ue, bytes, hipMemcpyHostToDevice, c->stream));
  else
    MALIO_HIP(hipMemcpy(c->d_unc, ue, bytes, hipMemcpyHostToDevice));
  for (int l = 0; l + 1 < L; l++) {
    for (int k = 0; k < 4; k++) c->tcq[l][k] = temporal_comp[l].q[k];
    for (int k = 0; k < 3; k++) c->tct[l][k] = temporal_comp[l].t[k];
  }
  return MALIO_OK;
}

// per-scan state that every new scan starts from. The per-point arrays (selection flags, neighbour cache, planes) are
// cleared by the kernel that sorts the scan at the first pass (k_gather_scan); nothing reads them before that.
static int scan_reset(Ctx *c) {
  c->node_guess_valid = false;  // malio_measure_node: a new scan starts with a plain (two-exchange) pass
  c->nbr_epoch = c->map_epoch;
  c->cert_valid = false;  // no search pass of this scan yet: nothing to keep (search_skip_begin)
  c->probe_valid = false;
  c->scan_sorted = false;
  c->last_M = -1;
  c->mm_guess_valid = false;  // the first pass of a scan runs as three kernels and leaves the first guess
  return MALIO_OK;
}

namespace malio {
// The caller's 48-byte points, already in HBM (copied there from page-locked memory by the DMA engine), packed to the
// 20-byte upload record; what the host loop of malio_scan_set does while it packs - counting the points of each LiDAR
// slot, validating the slot, noticing whether the slots come in ascending blocks - is left in info[] for the first pass:
// info[l] = points of slot l, info[8] = points with a slot outside [0, L), info[9] = descents of the slot sequence.
// info[0..9] and the scan's sequence number reach pinned memory from a LATER kernel of the same stream (the first
// workgroup of the grouping's first kernel, or k_publish_pack when the caller's order is to be kept and the counts are
// needed before that): the first pass picks them up there without a copy or a stream synchronisation
// (resolve_scan_segments), and the pack kernel does not end on a ticket and a write across PCIe.
__global__ void __launch_bounds__(BLK) k_pack_raw(const float *__restrict__ raw12, int n, int L, UploadRec *upload, u32 *info) {
  __shared__ u32 s_cnt[10];
  // The block's 256 points (12 KB) come in with fully coalesced 16-byte loads - every byte of the source is read exactly
  // once, which is what makes reading the caller's page-locked cloud in place (across PCIe, uncached) as fast as a DMA
  // copy of it - and are unpacked from LDS.
  __shared__ float4 s_raw[BLK * 3 + 1];
  if (threadIdx.x < 10) s_cnt[threadIdx.x] = 0;
  const int i0 = blockIdx.x * BLK;
  {
    const int nvec = min(BLK, n - i0) * 3;  // float4s of this block's points
    const float4 *src = reinterpret_cast<const float4 *>(raw12) + (size_t)i0 * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int e = threadIdx.x + k * BLK;
      if (e < nvec) s_raw[e] = src[e];
    }
    if (threadIdx.x == 0) s_raw[BLK * 3].x = i0 > 0 ? raw12[(size_t)(i0 - 1) * 12 + 8] : -1.f;  // the slot before this block's first point
  }
  __syncthreads();
  const int i = i0 + threadIdx.x;
  int lid = -1;  // -1: no point; MALIO_MAX_LIDAR: a slot outside [0, L)
  bool descent = false;
  if (i < n) {
    const float4 a = s_raw[threadIdx.x * 3];      // x y z _
    const float4 b = s_raw[threadIdx.x * 3 + 1];  // normal_x normal_y _ _
    lid = (int)s_raw[threadIdx.x * 3 + 2].x;      // intensity: the LiDAR slot, laserMapping.cpp:570
    int idx = (int)b.x;                           // int(laser_p.normal_x), laserMapping.cpp:694,737
    if (idx > 0x3FFFFF) idx = 0x3FFFFF;
    if (idx < -0x3FFFFF) idx = -0x3FFFFF;
    const float prev = threadIdx.x > 0 ? s_raw[threadIdx.x * 3 - 1].x : s_raw[BLK * 3].x;
    descent = i > 0 && (int)prev > lid;
    const bool bad = lid < 0 || lid >= L;
    UploadRec r;
    r.x = a.x, r.y = a.y, r.z = a.z;
    r.w = ((unsigned)idx << 8) | (unsigned)(bad ? 0 : lid);
    r.ny = b.y;
    upload[i] = r;
    if (bad) lid = MALIO_MAX_LIDAR;
  }
  // counted per wave (a wave's points nearly always share one slot: 64 same-address LDS atomics otherwise)
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int l = 0; l <= MALIO_MAX_LIDAR; l++) {
    const unsigned long long m = __ballot(lid == l);
    if (m && lane == 0) atomicAdd(&s_cnt[l == MALIO_MAX_LIDAR ? 8 : l], (u32)__popcll(m));
  }
  {
    const unsigned long long m = __ballot(descent);
    if (m && lane == 0) atomicAdd(&s_cnt[9], (u32)__popcll(m));
  }
  __syncthreads();
  if (threadIdx.x < 10 && s_cnt[threadIdx.x]) atomicAdd(&info[threadIdx.x], s_cnt[threadIdx.x]);
}
}  // namespace malio

// Was this very buffer copied ahead by malio_scan_stage? Consumes the record either way: a scan_set of anything else
// means the caller changed its mind, and a buffer staged twice must be copied twice.
static bool stage_take(Ctx *c, const void *buf, int n, int packed) {
  const bool hit = c->ahead_ptr && c->ahead_ptr == buf && c->ahead_n == n && c->ahead_packed == packed;
  c->ahead_ptr = nullptr;
  return hit;
}

// The NEXT scan's cloud on its way to HBM while the current scan is still being worked on - typically called right
// before malio_map_incremental, whose 0.15 ms hide the copy (90 us for 100 k 48-byte points, 38 us as 20-byte records):
// a copy stream of its own, a spare device buffer. The following malio_scan_set / malio_scan_set_packed of the SAME
// buffer and count uses the staged bytes and copies nothing. The buffer must be page-locked and stay untouched until
// malio_scan_upload_wait after that scan_set (the lifetime rule of malio_scan_set).
int malio_scan_stage(malio_handle_t h, const void *buf, int n, int packed) {
  if (check(h) || !buf || n <= 0) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  MALIO_HIP(hipSetDevice(c->device));
  const size_t bytes = (packed ? sizeof(UploadRec) : sizeof(malio_point_t)) * (size_t)n;
  hipPointerAttribute_t attr;
  bool pinned = hipPointerGetAttributes(&attr, buf) == hipSuccess && attr.type == hipMemoryTypeHost;
  if (pinned) pinned = hipPointerGetAttributes(&attr, static_cast<const char *>(buf) + bytes - 1) == hipSuccess && attr.type == hipMemoryTypeHost;
  (void)hipGetLastError();
  if (!pinned) {
    c->err = "malio_scan_stage: the buffer is not page-locked (malio_host_alloc / hipHostMalloc / hipHostRegister)";
    return MALIO_ERR_BAD_ARG;
  }
  if (!c->copy_stream) {
    MALIO_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    MALIO_HIP(hipEventCreateWithFlags(&c->ev_ahead, hipEventDisableTiming));
    MALIO_HIP(hipEventCreateWithFlags(&c->ev_ahead_free, hipEventDisableTiming));
  }
  if (c->ahead_busy) {  // the consumer of the previous staged cloud (pack kernel / device copy on `stream`) first
    MALIO_HIP(hipEventRecord(c->ev_ahead_free, c->stream));
    MALIO_HIP(hipStreamWaitEvent(c->copy_stream, c->ev_ahead_free, 0));
    c->ahead_busy = false;
  }
  if (bytes > c->cap_ahead) {
    MALIO_HIP(hipStreamSynchronize(c->copy_stream));
    MALIO_HIP(hipStreamSynchronize(c->stream));
    if (c->d_ahead) (void)hipFree(c->d_ahead);
    // (packed records: as large as the upload array, so that malio_scan_set_packed can swap the two instead of copying)
    c->d_ahead = nullptr, c->cap_ahead = std::max(bytes + bytes / 8 + 4096, sizeof(UploadRec) * c->cap_scan);
    MALIO_HIP(hipMalloc(&c->d_ahead, c->cap_ahead));
  }
  MALIO_HIP(hipMemcpyAsync(c->d_ahead, buf, bytes, hipMemcpyHostToDevice, c->copy_stream));
  MALIO_HIP(hipEventRecord(c->ev_ahead, c->copy_stream));
  c->ahead_ptr = buf, c->ahead_n = n, c->ahead_packed = packed ? 1 : 0;
  return MALIO_OK;
}

int malio_scan_set(malio_handle_t h, const malio_point_t *body, int n, const malio_pose_t *const *pose_unc,
                   const int *pose_unc_len, const malio_pose_t *temporal_comp) {
  if (check(h) || !body || n <= 0 || !pose_unc || !pose_unc_len) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  const int L = c->prm.lid_num;
  if (L > 1 && !temporal_comp) return MALIO_ERR_BAD_ARG;
  MALIO_HIP(hipSetDevice(c->device));
  for (int l = 0; l < L; l++)
    if (pose_unc_len[l] < 2 || !pose_unc[l]) return MALIO_ERR_BAD_ARG;  // (before anything is queued; scan_tables repeats it)
  // ONE pass over the caller's cloud: pack to 20 B in the caller's order into pinned memory, count the points of each
  // LiDAR slot; one copy into HBM that nothing here waits for. The scan sort groups the slots (its key leads with it).
  c->N = n;
  int rc = measure_alloc(c);
  if (rc != MALIO_OK) return rc;
  c->seg_pending = false;
  // what a previous scan that never reached its first pass (dropped frame, MALIO_ERR_NO_MAP) may have left armed: its
  // counts were to be formed by the grouping's first kernel (malio_scan_set_packed) - this scan brings its own
  c->count_in_sort = false, c->pack_publish_pending = false;
  {
    // A cloud in page-locked memory (malio_host_alloc, or any hipHostMalloc / hipHostRegister'ed buffer) is not touched
    // by this thread at all: one DMA copy of the 48-byte points and a kernel that packs them; the per-slot counts come
    // back with the first pass (resolve_scan_segments). A pageable cloud would be staged by the runtime page by page
    // (~0.7 ms per 10 MB): it is packed here instead, in one pass over it.
    const bool staged = stage_take(c, body, n, 0);  // copied ahead by malio_scan_stage: no copy at all here
    hipPointerAttribute_t attr;
    bool pinned = staged || (hipPointerGetAttributes(&attr, body) == hipSuccess && attr.type == hipMemoryTypeHost);
    if (pinned && !staged) {  // ... and the last byte as well: a cloud that merely starts inside a registered range is staged like any other
      const char *last = reinterpret_cast<const char *>(body) + sizeof(malio_point_t) * (size_t)n - 1;
      pinned = hipPointerGetAttributes(&attr, last) == hipSuccess && attr.type == hipMemoryTypeHost;
    }
    (void)hipGetLastError();
    if (pinned) {
      // (the pack kernel reading the page-locked cloud in place instead - no copy call, which costs this thread ~16 us -
      // was measured: 109 us of kernel against 90 us of DMA + 15 us of pack, the turn 10 us slower)
      if (staged) {
        MALIO_HIP(hipStreamWaitEvent(c->stream, c->ev_ahead, 0));
      } else {
        if ((size_t)n > c->cap_raw) {
          if (c->d_raw) (void)hipFree(c->d_raw);
          c->d_raw = nullptr, c->cap_raw = (size_t)n + (size_t)n / 8 + 1024;
          MALIO_HIP(hipMalloc(&c->d_raw, sizeof(float) * 12 * c->cap_raw));
        }
        MALIO_HIP(hipMemcpyAsync(c->d_raw, body, sizeof(float) * 12 * (size_t)n, hipMemcpyHostToDevice, c->stream));
      }
      // the caller's buffer is in use until this event (include/malio.h: lifetime of feats_down_body)
      if (!c->ev_upload) MALIO_HIP(hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming));
      MALIO_HIP(hipEventRecord(c->ev_upload, c->stream));
      c->upload_in_flight = true;
      if (c->scan_set_sync) {  // MALIO_OPT_SCAN_SET_SYNC
        MALIO_HIP(hipEventSynchronize(c->ev_upload));
        c->upload_in_flight = false;
      }
      const float *src = staged ? static_cast<const float *>(c->d_ahead) : c->d_raw;
      if (!c->d_packinfo) MALIO_HIP(hipMalloc(&c->d_packinfo, sizeof(u32) * 16));
      MALIO_HIP(hipMemsetAsync(c->d_packinfo, 0, sizeof(u32) * 16, c->stream));
      c->packinfo_clean = false;  // (k_pack_raw leaves its counts in it: a later malio_scan_set_packed must clear them)
      if (!c->h_packinfo) {
        MALIO_HIP(hipHostMalloc((void **)&c->h_packinfo, sizeof(u32) * 16, hipHostMallocMapped | hipHostMallocCoherent));
        MALIO_HIP(hipHostGetDevicePointer((void **)&c->d_packinfo_pub, c->h_packinfo, 0));
        memset(c->h_packinfo, 0, sizeof(u32) * 16);
      }
      if (++c->pack_seq == 0) c->pack_seq = 1;
      hipLaunchKernelGGL(k_pack_raw, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, c->stream, src, n, L, c->d_upload, c->d_packinfo);
      if (staged) c->ahead_busy = true;  // (the next malio_scan_stage waits for this kernel before it overwrites d_ahead)
      c->pack_publish_pending = true;
      if (c->scan_order_mode == MALIO_SCAN_ORDER_KEEP) publish_pack_now(c);
      MALIO_HIP(hipGetLastError());
      c->seg_pending = true;
      c->scan_keep_order = false;  // decided when the counts arrive
      // the uncertainty tables are folded (tens of us of host arithmetic) while the cloud is on its way
      if (int rct = scan_tables(c, pose_unc, pose_unc_len, temporal_comp)) {
        c->N = 0, c->seg_pending = false;
        return rct;
      }
      return scan_reset(c);
    }
  }
  if (int rct = scan_tables(c, pose_unc, pose_unc_len, temporal_comp)) {
    c->N = 0;
    return rct;
  }
  UploadRec *stage = nullptr;
  if (int rcs = host_stage(c, sizeof(UploadRec) * (size_t)n, (void **)&stage)) return rcs;
  int cnt[MALIO_MAX_LIDAR] = {0};
  int last_lid = 0;
  bool grouped = true;  // LiDAR slots in ascending blocks: what MALIO_SCAN_ORDER_KEEP needs
  for (int i = 0; i < n; i++) {
    const int lid = (int)body[i].intensity;  // laserMapping.cpp:570
    if (lid < 0 || lid >= L) {
      c->N = 0;
      return MALIO_ERR_BAD_ARG;
    }
    grouped &= lid >= last_lid;
    last_lid = lid;
    int idx = (int)body[i].normal_x;  // int(laser_p.normal_x), laserMapping.cpp:694,737
    if (idx > 0x3FFFFF) idx = 0x3FFFFF;
    if (idx < -0x3FFFFF) idx = -0x3FFFFF;
    cnt[lid]++;
    UploadRec &r = stage[i];
    r.x = body[i].x, r.y = body[i].y, r.z = body[i].z;
    r.w = ((unsigned)idx << 8) | (unsigned)lid;
    r.ny = body[i].normal_y;
  }
  c->seg_start[0] = 0;
  for (int l = 0; l < MALIO_MAX_LIDAR; l++) c->seg_start[l + 1] = c->seg_start[l] + (l < L ? cnt[l] : 0);
  MALIO_HIP(hipMemcpyAsync(c->d_upload, stage, sizeof(UploadRec) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  c->stage_pending = true;
  c->scan_keep_order = c->scan_order_mode == MALIO_SCAN_ORDER_KEEP && grouped;  // not grouped: sorted after all
  return scan_reset(c);
}

// ---- resident front end: voxel filter + scan upload straight from the undistorted clouds in HBM -------------------
namespace malio {
// one down-sampled LiDAR cloud -> its segment of the scan: the field shuffle of laserMapping.cpp:972-976
// (normal_x <- intensity [the voxel mean of the uncertainty index], intensity <- LiDAR number) and the 16-byte pack
__global__ void __launch_bounds__(BLK) k_pack_resident(const float *__restrict__ down12, int n, int lid, int dst0,
                                                       UploadRec *upload, float *body12) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  const float *p = down12 + (size_t)i * 12;
  int idx = (int)p[8];  // int(laser_p.normal_x), laserMapping.cpp:694,737
  if (idx > 0x3FFFFF) idx = 0x3FFFFF;
  if (idx < -0x3FFFFF) idx = -0x3FFFFF;
  UploadRec r;
  r.x = p[0], r.y = p[1], r.z = p[2];
  r.w = ((unsigned)idx << 8) | (unsigned)lid;
  r.ny = p[5];
  upload[dst0 + i] = r;
  if (body12) {
    float *q = body12 + (size_t)(dst0 + i) * 12;
#pragma unroll
    for (int k = 0; k < 12; k++) q[k] = p[k];
    q[4] = p[8];
    q[8] = (float)lid;
  }
}
}  // namespace malio

namespace malio {
// what k_pack_raw leaves in info[] for records that arrive packed: points per LiDAR slot, slots outside [0, L), descents of
// the slot sequence
__global__ void __launch_bounds__(BLK) k_count_packed(UploadRec *rec, int n, int L, u32 *info) {
  __shared__ u32 s_cnt[10];
  if (threadIdx.x < 10) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int i = blockIdx.x * BLK + threadIdx.x;
  int lid = -1;
  bool descent = false;
  if (i < n) {
    const u32 w = rec[i].w;
    lid = (int)(w & 0xFFu);
    descent = i > 0 && (int)(rec[i - 1].w & 0xFFu) > lid;  // (a neighbour's bad slot may already read 0: the scan is refused anyway)
    if (lid >= L) {  // counted, reported by the first pass; slot 0 meanwhile, so that the grouping stays inside its buckets
      lid = MALIO_MAX_LIDAR;
      rec[i].w = w & 0xFFFFFF00u;
    }
  }
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int l = 0; l <= MALIO_MAX_LIDAR; l++) {
    const unsigned long long m = __ballot(lid == l);
    if (m && lane == 0) atomicAdd(&s_cnt[l == MALIO_MAX_LIDAR ? 8 : l], (u32)__popcll(m));
  }
  {
    const unsigned long long m = __ballot(descent);
    if (m && lane == 0) atomicAdd(&s_cnt[9], (u32)__popcll(m));
  }
  __syncthreads();
  if (threadIdx.x < 10 && s_cnt[threadIdx.x]) atomicAdd(&info[threadIdx.x], s_cnt[threadIdx.x]);
}
}  // namespace malio

int malio_scan_set_packed(malio_handle_t h, const malio_scan_rec_t *recs, int n, const malio_pose_t *const *pose_unc,
                          const int *pose_unc_len, const malio_pose_t *temporal_comp) {
  if (check(h) || !recs || n <= 0 || !pose_unc || !pose_unc_len) return MALIO_ERR_BAD_ARG;
  static_assert(sizeof(malio_scan_rec_t) == sizeof(UploadRec), "malio_scan_rec_t is the upload record");
  Ctx *c = h;
  const int L = c->prm.lid_num;
  if (L > 1 && !temporal_comp) return MALIO_ERR_BAD_ARG;
  MALIO_HIP(hipSetDevice(c->device));
  for (int l = 0; l < L; l++)
    if (pose_unc_len[l] < 2 || !pose_unc[l]) return MALIO_ERR_BAD_ARG;
  c->N = n;
  int rc = measure_alloc(c);
  if (rc != MALIO_OK) return rc;
  c->seg_pending = false;
  const bool staged = stage_take(c, recs, n, 1);  // copied ahead by malio_scan_stage: a device-to-device copy is left
  hipPointerAttribute_t attr;
  bool pinned = staged || (hipPointerGetAttributes(&attr, recs) == hipSuccess && attr.type == hipMemoryTypeHost);
  if (pinned && !staged) {
    const char *last = reinterpret_cast<const char *>(recs) + sizeof(malio_scan_rec_t) * (size_t)n - 1;
    pinned = hipPointerGetAttributes(&attr, last) == hipSuccess && attr.type == hipMemoryTypeHost;
  }
  (void)hipGetLastError();
  const void *src = recs;
  if (!pinned) {  // through the handle's staging buffer: one pass over the records, one copy nothing here waits for
    void *stage = nullptr;
    if (int rcs = host_stage(c, sizeof(UploadRec) * (size_t)n, &stage)) return rcs;
    memcpy(stage, recs, sizeof(UploadRec) * (size_t)n);
    src = stage;
    c->stage_pending = true;
  }
  if (staged) {
    MALIO_HIP(hipStreamWaitEvent(c->stream, c->ev_ahead, 0));
    if (c->cap_ahead >= sizeof(UploadRec) * c->cap_scan) {
      // the staged buffer BECOMES the upload array (what was the upload array is the next scan's staging buffer):
      // nothing is copied
      void *old_up = c->d_upload;
      c->d_upload = static_cast<UploadRec *>(c->d_ahead);
      c->d_ahead = old_up, c->cap_ahead = sizeof(UploadRec) * c->cap_scan;
    } else {
      MALIO_HIP(hipMemcpyAsync(c->d_upload, c->d_ahead, sizeof(UploadRec) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
    }
    c->ahead_busy = true;
  } else {
    MALIO_HIP(hipMemcpyAsync(c->d_upload, src, sizeof(UploadRec) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  }
  if (pinned) {
    if (!c->ev_upload) MALIO_HIP(hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming));
    MALIO_HIP(hipEventRecord(c->ev_upload, c->stream));
    c->upload_in_flight = true;
  }
  // the per-slot counts (and the validation of the slots) come back with the first pass, as for a page-locked cloud.
  // A scan that will be grouped is counted by the grouping's first kernel (count_in_sort): no kernel and no clear here.
  const bool cis = c->scan_order_mode != MALIO_SCAN_ORDER_KEEP;
  const bool first_info = !c->d_packinfo;
  if (!c->d_packinfo) MALIO_HIP(hipMalloc(&c->d_packinfo, sizeof(u32) * 16));
  if (!cis || first_info || !c->packinfo_clean) MALIO_HIP(hipMemsetAsync(c->d_packinfo, 0, sizeof(u32) * 16, c->stream));
  c->packinfo_clean = cis;  // (k_sort_scan leaves it cleared; every other producer leaves its counts in it)
  if (!c->h_packinfo) {
    MALIO_HIP(hipHostMalloc((void **)&c->h_packinfo, sizeof(u32) * 16, hipHostMallocMapped | hipHostMallocCoherent));
    MALIO_HIP(hipHostGetDevicePointer((void **)&c->d_packinfo_pub, c->h_packinfo, 0));
    memset(c->h_packinfo, 0, sizeof(u32) * 16);
  }
  if (++c->pack_seq == 0) c->pack_seq = 1;
  if (!cis) hipLaunchKernelGGL(k_count_packed, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, c->stream, c->d_upload, n, L, c->d_packinfo);
  c->count_in_sort = cis;
  c->pack_publish_pending = true;
  if (c->scan_order_mode == MALIO_SCAN_ORDER_KEEP) publish_pack_now(c);
  MALIO_HIP(hipGetLastError());
  c->seg_pending = true;
  c->scan_keep_order = false;
  if (int rct = scan_tables(c, pose_unc, pose_unc_len, temporal_comp)) {
    c->N = 0, c->seg_pending = false;
    return rct;
  }
  return scan_reset(c);
}

int malio_scan_upload_wait(malio_handle_t h) {
  if (check(h)) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  if (!c->upload_in_flight || !c->ev_upload) return MALIO_OK;
  MALIO_HIP(hipSetDevice(c->device));
  MALIO_HIP(hipEventSynchronize(c->ev_upload));
  c->upload_in_flight = false;
  return MALIO_OK;
}

int malio_scan_set_resident(malio_handle_t h, float leaf, int normal_mode, const malio_pose_t *const *pose_unc,
                            const int *pose_unc_len, const malio_pose_t *temporal_comp, malio_point_t *out_body, int cap,
                            int *out_n) {
  if (check(h) || !pose_unc || !pose_unc_len || !out_n || cap < 0 || (cap > 0 && !out_body) || !(leaf > 0.f)) return MALIO_ERR_BAD_ARG;
  if (normal_mode != MALIO_VOXEL_NORMAL_MEAN && normal_mode != MALIO_VOXEL_NORMAL_NORMALIZE) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  const int L = c->prm.lid_num;
  if (L > 1 && !temporal_comp) return MALIO_ERR_BAD_ARG;
  MALIO_HIP(hipSetDevice(c->device));
  if (int rct = scan_tables(c, pose_unc, pose_unc_len, temporal_comp)) return rct;
  ArenaScope sc(c->arena);
  const float *down[MALIO_MAX_LIDAR] = {nullptr};
  int m[MALIO_MAX_LIDAR] = {0};
  int n = 0;
  for (int l = 0; l < L; l++) {  // downSizeFilterSurf per LiDAR (:968-971), clouds concatenated in LiDAR order (:982)
    if (c->res[l].n <= 0) continue;
    float *d = nullptr;
    bool pass = false;
    int rc = voxel_downsample_dev(c, sc, c->res[l].d, c->res[l].n, leaf, normal_mode, &d, &m[l], &pass);
    if (rc != MALIO_OK) return rc;
    down[l] = pass ? c->res[l].d : d;
    n += m[l];
  }
  *out_n = n;
  if (n <= 0) return MALIO_ERR_NO_SCAN;
  c->N = n;
  c->seg_pending = false;  // (a page-locked malio_scan_set nobody ran a pass on may have left its counts pending,
  c->count_in_sort = false, c->pack_publish_pending = false;  // a packed one its counting armed for the grouping)
  c->seg_start[0] = 0;
  for (int l = 0; l < MALIO_MAX_LIDAR; l++) c->seg_start[l + 1] = c->seg_start[l] + (l < L ? m[l] : 0);
  int rc = measure_alloc(c);
  if (rc != MALIO_OK) return rc;
  float *d_body = nullptr;
  const bool want_body = out_body && cap > 0;
  if (want_body) MALIO_HIP(sc.get(&d_body, (size_t)n * 12));
  for (int l = 0; l < L; l++)
    if (m[l] > 0)
      hipLaunchKernelGGL(k_pack_resident, dim3((m[l] + BLK - 1) / BLK), dim3(BLK), 0, c->stream, down[l], m[l], l,
                         c->seg_start[l], c->d_upload, d_body);
  if (want_body) {
    MALIO_HIP(hipMemcpyAsync(out_body, d_body, sizeof(float) * 12 * (size_t)std::min(n, cap), hipMemcpyDeviceToHost, c->stream));
    MALIO_HIP(hipStreamSynchronize(c->stream));  // out_body is the caller's memory
  }
  rc = scan_reset(c);
  // the voxel filter leaves every LiDAR's cloud sorted by voxel index and the clouds follow each other in slot order:
  // spatially coherent and grouped as it is
  c->scan_keep_order = c->scan_order_mode != MALIO_SCAN_ORDER_SORT;
  for (int l = 0; l < L; l++) c->res[l].n = 0;  // consumed
  return rc;
}

int malio_sums_len(malio_handle_t h) { return h ? sums_len(h) : 0; }

int malio_measure_stage1(malio_handle_t h, const malio_state_t *s, int converge, double *d_minmax4) {
  if (check(h) || !s) return MALIO_ERR_BAD_ARG;
  MALIO_HIP_H(hipSetDevice(h->device));
  prof_begin(h);
  return pass_stage1(h, s, converge, d_minmax4);  // NULL: the extrema are emitted by malio_measure_stage2_emit
}
int malio_measure_stage2(malio_handle_t h, const double *d_minmax4, double *d_sums) {
  if (check(h) || !d_minmax4 || !d_sums) return MALIO_ERR_BAD_ARG;
  int rc = pass_stage2(h, d_minmax4, nullptr, d_sums, false);
  return rc;
}
int malio_measure_stage2_emit(malio_handle_t h, const double *d_minmax4_in, double *d_minmax_out, double *d_sums) {
  if (check(h) || !d_minmax4_in || !d_minmax_out || !d_sums) return MALIO_ERR_BAD_ARG;
  return pass_stage2(h, d_minmax4_in, d_minmax_out, d_sums, false);
}
int malio_measure_node(malio_handle_t h, malio_xchg_t x, const malio_state_t *s, int converge, malio_measure_out_t *out,
                       int *stats2) {
  if (check(h) || !x || !s || !out) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  MALIO_HIP(hipSetDevice(c->device));
  if (c->N <= 0) return MALIO_ERR_NO_SCAN;
  const int ns = sums_len(c);
  if (malio_xchg_row(x) != ns + MALIO_MINMAX_LEN) return MALIO_ERR_BAD_ARG;
  if (!c->d_node_mm) {
    MALIO_HIP(hipMalloc(&c->d_node_mm, sizeof(double) * MALIO_MINMAX_LEN));
    MALIO_HIP(hipHostMalloc(&c->h_node_mm, sizeof(double) * MALIO_MINMAX_LEN, hipHostMallocDefault));
  }
  const double timeout_s = 60.0;
  double *res = c->h_res;  // [sums | extrema words]: written by the kernels, exchanged, reduced in place
  auto upload = [&](const double *E) -> int {  // the extrema stage 2 weights the rows with; only when they change
    if (c->node_uploaded_valid && memcmp(E, c->node_uploaded, sizeof(double) * 4) == 0) return MALIO_OK;
    memcpy(c->h_node_mm, E, sizeof(double) * 4);
    MALIO_HIP(hipMemcpyAsync(c->d_node_mm, c->h_node_mm, sizeof(double) * 4, hipMemcpyHostToDevice, c->stream));
    memcpy(c->node_uploaded, E, sizeof(double) * 4);
    c->node_uploaded_valid = true;
    return MALIO_OK;
  };
  prof_begin(c);
  // where the kernels leave [sums | extrema words]: the handle's pinned result buffer (the host exchanges it through
  // shared memory), or the exchange's device row (RCCL gathers it on the stream, right behind the kernels)
  double *row = c->d_res;
  const bool dev_xchg = malio_xchg_kind(x) == 2;
  if (dev_xchg && (rc_dev_row(x, &row) != MALIO_OK)) return MALIO_ERR_BAD_ARG;
  double own[MALIO_MINMAX_LEN] = {0};
  // host exchange: the kernel that ends a stage announces it through a sequence word in pinned memory (as malio_measure's
  // passes do) and this thread polls it - the row is in hand ~5 us before a stream synchronisation would have returned
  volatile int *h_msg = nullptr;
  GateArgs gate{};
  const bool poll = !dev_xchg && !c->profiling;
  auto arm = [&]() -> const GateArgs * {
    if (!poll) return nullptr;
    int *d_msg = nullptr;
    if (gate_words(c, &h_msg, &d_msg)) return nullptr;
    c->gate_epoch = c->gate_epoch >= (1 << 30) ? 1 : c->gate_epoch + 1;
    gate = GateArgs{};
    gate.msg_seq = d_msg, gate.ticket = c->d_gate_ticket, gate.publish = c->gate_epoch;
    return &gate;
  };
  auto reduce = [&](const double *guess, double *Eout) -> int {  // one exchange: gather, true extrema, rank-ordered sum
    if (dev_xchg) return malio_xchg_reduce_stream(x, c->stream, ns, guess, res, Eout, own + 4);
    if (poll && gate.msg_seq) {
      long long spins = 0;
      while (__atomic_load_n(const_cast<int *>(h_msg), __ATOMIC_ACQUIRE) != gate.publish) {
        if ((++spins & 0xFFFF) == 0 && hipStreamQuery(c->stream) != hipErrorNotReady) {
          if (__atomic_load_n(const_cast<int *>(h_msg), __ATOMIC_ACQUIRE) == gate.publish) break;
          MALIO_HIP(hipStreamSynchronize(c->stream));  // surfaces the error that ended the queue early
          c->err = "malio_measure_node: the pass ended without announcing its result";
          return MALIO_ERR_HIP;
        }
        __builtin_ia32_pause();
      }
      gate.msg_seq = nullptr;
    } else {
      MALIO_HIP(hipStreamSynchronize(c->stream));
    }
    memcpy(own + 4, res + ns + 4, sizeof(double) * 4);
    // word 5 of the extrema words (reserved, 0 from the kernels): which update loop this shard is in. Every shard of a node
    // must run the same one - the gated chain and the pass-by-pass loop meet in different exchanges - and each decides it
    // from its own handle's options: a disagreement is caught HERE, in an exchange both loops go through, by every rank
    // at once, instead of as a 60 s time-out in the next one.
    res[ns + 5] = c->node_mode_word;
    const int rcx = malio_xchg_reduce(x, res, ns, guess, res, Eout, timeout_s);
    if (rcx >= 0 && !xchg_word_agrees(x, ns + 5)) {
      c->err = "malio_measure_node: the shards of this node run different update loops (gated chain on some, pass by pass on "
               "others): MALIO_OPT_NODE_GATED, MALIO_OPT_FUSE, the update mode, a pass hook and profiling must be the same on every shard";
      return MALIO_ERR_BAD_ARG;
    }
    return rcx;
  };
  const bool spec = c->node_guess_valid;
  // A speculating pass is ONE kernel + the final sum where the single-GPU pass is (k_pass: the rows are weighted with the
  // previous pass' GLOBAL extrema; workgroups of other shards' tiles contribute zero tiles); hit or miss is decided
  // across the shards by the exchange below, a miss redoes the rows from the per-point state as after k_search.
  const bool fused = spec && fuse_eligible(c, converge, false);
  int rc = MALIO_OK;
  double E[4];
  bool have_sums = false;
  if (fused) {
    memcpy(c->mm_guess, c->node_guess, sizeof(double) * 4);
    c->mm_guess_valid = true;
    if ((rc = pass_fused(c, s, converge, arm(), row)) != MALIO_OK) return rc;
    rc = reduce(c->node_guess, E);
    if (rc < 0) return rc;
    have_sums = rc == MALIO_OK;
    fuse_note(c, have_sums);
    if (have_sums) c->node_hits++;
  } else if ((rc = pass_stage1(c, s, converge, spec ? nullptr : row + ns)) != MALIO_OK) {
    return rc;
  } else if (spec) {
    if ((rc = upload(c->node_guess)) != MALIO_OK) return rc;
    if ((rc = pass_stage2(c, c->d_node_mm, row + ns, row, false, arm())) != MALIO_OK) return rc;
    rc = reduce(c->node_guess, E);
    if (rc < 0) return rc;
    have_sums = rc == MALIO_OK;
    if (have_sums) c->node_hits++;
  } else {
    const double nan4[4] = {NAN, NAN, NAN, NAN};  // never equal: only the extrema are wanted from this exchange
    rc = reduce(nan4, E);
    if (rc < 0) return rc;
  }
  if (!have_sums) {  // first pass of a scan, or the extrema moved: weight the rows with the true extrema
    if (spec) c->node_misses++;
    double keep[4];
    memcpy(keep, own + 4, sizeof(keep));  // (this rank's own words come with the extrema: the second round has none)
    if ((rc = upload(E)) != MALIO_OK) return rc;
    if ((rc = pass_stage2(c, c->d_node_mm, nullptr, row, false, arm())) != MALIO_OK) return rc;
    double E2[4];
    rc = reduce(nullptr, E2);
    if (rc != MALIO_OK) return rc < 0 ? rc : MALIO_ERR_HIP;
    memcpy(own + 4, keep, sizeof(keep));
  }
  prof_end(c);
  memcpy(c->node_guess, E, sizeof(E));
  c->node_guess_valid = true;
  memcpy(res + ns, E, sizeof(E));  // finish_host reads the global extrema here, then this rank's own words
  memcpy(res + ns + 4, own + 4, sizeof(double) * 4);
  rc = finish_host(c, res, res + ns, out);
  c->last_M = out->M;
  if (stats2) stats2[0] = c->node_hits, stats2[1] = c->node_misses;
  return rc;
}

int malio_update_iterated_node(malio_handle_t h, malio_xchg_t xchg, malio_state_t *x, double *P, double R, int *stats,
                               double *solve_time) {
  if (check(h) || !xchg || !x || !P) return MALIO_ERR_BAD_ARG;
  MALIO_HIP_H(hipSetDevice(h->device));
  if (solve_time) *solve_time = 0;
  // The gated chain where it can run: rows exchanged through host memory (RCCL's gather is a stream operation: the gate
  // would wait behind it), the one-kernel pass allowed, nobody watching single passes. Every shard of a node must decide
  // this the same way: the options are the node's (malio_node_set_option), a pass hook keeps the node's loop on one thread.
  if (h->opt_node_gated && h->update_mode == MALIO_UPDATE_GATED && malio_xchg_kind(xchg) != 2 && h->fuse_enabled && !h->pass_hook &&
      !h->profiling && h->prm.max_iteration >= 1) {
    malio_state_t x0 = *x;
    h->node_mode_word = 1.0;
    const int rc = ieskf_update_gated(h, xchg, x, P, stats, solve_time);
    h->node_mode_word = 0.0;
    h->node_gated_runs++;
    if (rc != MALIO_SMALL_M_FALLBACK) return rc;
    *x = x0;  // (untouched by contract; the per-pass loop below redoes the update - and reports M < n itself)
    h->node_gated_redone++;
    // (every shard left the chain together - the poison row - and redoes the update here: mode 1 again, so that a node whose
    // shards all fell back does not look like a disagreement)
    h->node_mode_word = 1.0;
    const int rc2 = ieskf_update(h, xchg, x, P, R, stats, solve_time);
    h->node_mode_word = 0.0;
    return rc2;
  }
  h->node_mode_word = 2.0;
  const int rc = ieskf_update(h, xchg, x, P, R, stats, solve_time);
  h->node_mode_word = 0.0;
  return rc;
}

int malio_node_stats(malio_handle_t h, int *stats2) {
  if (check(h) || !stats2) return MALIO_ERR_BAD_ARG;
  stats2[0] = h->node_hits, stats2[1] = h->node_misses;
  return MALIO_OK;
}

int malio_measure_finish(malio_handle_t h, const double *sums_host, const double *minmax_host,
                         malio_measure_out_t *out) {
  if (check(h) || !sums_host || !minmax_host || !out) return MALIO_ERR_BAD_ARG;
  prof_end(h);
  int rc = finish_host(h, sums_host, minmax_host, out);
  h->last_M = out->M;
  return rc;
}

int malio_measure(malio_handle_t h, const malio_state_t *s, int converge, malio_measure_out_t *out) {
  if (check(h) || !s || !out) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  MALIO_HIP(hipSetDevice(c->device));
  if (c->N <= 0) return MALIO_ERR_NO_SCAN;
  const bool want_rows = out->h_x || out->h || out->R;
  prof_begin(c);
  const int ns = sums_len(c);
  // The two result kernels store straight into pinned, device-mapped host memory (2.4 KB over PCIe): no copy kernel
  // between the last kernel and the host (worth ~1 us per pass).
  // How the call learns that the pass is over: the last workgroup of the last kernel stores a sequence word into pinned
  // memory (after the sums, which go there too) and this thread polls it - the results are in hand ~2 us before the
  // queue's completion signal would have woken a hipStreamSynchronize. (Profiling and the rows path keep the sync.)
  volatile int *h_msg = nullptr;
  GateArgs gate{};
  const bool poll = !c->profiling && !want_rows;
  auto arm = [&]() -> int {  // a fresh sequence number for the kernel that ends the pass
    if (!poll) return MALIO_OK;
    int *d_msg = nullptr;
    if (int rcg = gate_words(c, &h_msg, &d_msg)) return rcg;
    c->gate_epoch = c->gate_epoch >= (1 << 30) ? 1 : c->gate_epoch + 1;
    gate.msg_seq = d_msg, gate.ticket = c->d_gate_ticket, gate.publish = c->gate_epoch;
    return MALIO_OK;
  };
  auto wait = [&]() -> int {
    if (!poll) {
      MALIO_HIP(hipStreamSynchronize(c->stream));
      return MALIO_OK;
    }
    long long spins = 0;
    while (__atomic_load_n(const_cast<int *>(h_msg), __ATOMIC_ACQUIRE) != gate.publish) {
      if ((++spins & 0xFFFF) == 0 && hipStreamQuery(c->stream) != hipErrorNotReady) {
        if (__atomic_load_n(const_cast<int *>(h_msg), __ATOMIC_ACQUIRE) == gate.publish) break;
        MALIO_HIP(hipStreamSynchronize(c->stream));  // surfaces the error that ended the queue early
        c->err = "malio_measure: the pass ended without announcing its result";
        return MALIO_ERR_HIP;
      }
      __builtin_ia32_pause();
    }
    return MALIO_OK;
  };
  int rc;
  const double *res = c->h_res;
  bool need_stage2 = true;
  if (!want_rows && fuse_eligible(c, converge)) {  // (round 6: reuse passes too - k_reuse_rows, the streaming form - not only search passes)
    // k_pass -> k_final_reduce: the rows are formed inside the point-phase kernel, weighted with the extrema of the
    // previous pass of this scan; the host checks the guess. A miss (rare: an extreme point changed sides) costs the two
    // kernels below.
    if ((rc = arm()) != MALIO_OK) return rc;
    if (!poll) gate.msg_seq = nullptr;
    if ((rc = pass_fused(c, s, converge, poll ? &gate : nullptr)) != MALIO_OK) return rc;
    if ((rc = wait()) != MALIO_OK) return rc;
    bool hit = false;
    fused_collect(c, nullptr, &hit);
    if (hit) need_stage2 = false;
  } else {
    rc = pass_stage1(c, s, converge, nullptr);
    if (rc != MALIO_OK) return rc;
  }
  if (need_stage2) {
    if ((rc = arm()) != MALIO_OK) return rc;
    rc = pass_stage2(c, nullptr, c->d_res + ns, c->d_res, want_rows, poll ? &gate : nullptr);
    if (rc != MALIO_OK) return rc;
    if ((rc = wait()) != MALIO_OK) return rc;
    res = c->h_res;
  }
  prof_end(c);
  rc = finish_host(c, res, res + ns, out);
  c->last_M = out->M;
  memcpy(c->mm_guess, res + ns, sizeof(double) * 4);  // the next pass of this scan speculates on these
  c->mm_guess_valid = true;
  if (want_rows && out->valid) {
    // Rows path (parity tests, M < n fallback): dense per-point rows back to the host, expanded to
    // C columns, scaled by w_loc (laserMapping.cpp:758-759), compacted in ascending original index.
    const int N = c->N, L = c->prm.lid_num, C = 6 * (1 + L);
    std::vector<double> rows((size_t)N * 14);
    std::vector<u32> perm(N);
    std::vector<unsigned char> sel(N);
    MALIO_HIP(hipMemcpy(rows.data(), c->d_rows, sizeof(double) * rows.size(), hipMemcpyDeviceToHost));
    MALIO_HIP(hipMemcpy(perm.data(), c->d_perm, sizeof(u32) * N, hipMemcpyDeviceToHost));
    MALIO_HIP(hipMemcpy(sel.data(), c->d_sel, N, hipMemcpyDeviceToHost));
    std::vector<int> src_of(N, -1);  // original index -> sorted index
    for (int i = 0; i < N; i++) src_of[perm[i]] = i;
    int m = 0;
    for (int o = 0; o < N; o++) {
      int i = src_of[o];
      if (i < 0 || !sel[i]) continue;
      int lid = 0;
      for (int l = 1; l < L; l++)
        if (i >= c->seg_start[l]) lid = l;
      const double *r = &rows[(size_t)i * 14];
      if (out->h_x) {
        double *dst = out->h_x + (size_t)m * C;
        for (int k = 0; k < C; k++) dst[k] = 0;
        for (int k = 0; k < 6; k++) dst[k] = r[k] * out->w_loc;
        for (int k = 0; k < 3; k++) {
          dst[6 + 3 * lid + k] = r[6 + k] * out->w_loc;
          dst[6 + 3 * (L + lid) + k] = r[9 + k] * out->w_loc;
        }
      }
      if (out->h) out->h[m] = r[12] * out->w_loc;
      if (out->R) out->R[m] = r[13];
      m++;
    }
  }
  return rc;
}

int malio_scan_get(malio_handle_t h, float *normal_y, malio_point_t *nearest, int *nearest_count, uint8_t *selected,
                   float *res_last, float *world_xyz, float *normvec4) {
  if (check(h)) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  if (c->N <= 0 || !c->scan_sorted) return MALIO_ERR_NO_SCAN;
  MALIO_HIP(hipSetDevice(c->device));
  const int N = c->N;
  std::vector<u32> perm(N);
  std::vector<unsigned char> sel(N), nf(N);
  std::vector<double> tr(N);
  std::vector<float> ny(N);
  std::vector<float> pd2(N), world((size_t)3 * N);
  std::vector<float4> world4;  // feats_down_world after a search pass (the search kernel keeps one copy of it)
  const bool from_search = c->last_pass_search;
  std::vector<float4> plane(N);
  MALIO_HIP(hipMemcpyAsync(perm.data(), c->d_perm, sizeof(u32) * N, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipMemcpyAsync(sel.data(), c->d_sel, N, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipMemcpyAsync(nf.data(), c->d_nfound, N, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipMemcpyAsync(tr.data(), c->d_trace, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipMemcpyAsync(ny.data(), c->d_ny, sizeof(float) * N, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipMemcpyAsync(pd2.data(), c->d_pd2, sizeof(float) * N, hipMemcpyDeviceToHost, c->stream));
  if (from_search) {
    world4.resize(N);
    MALIO_HIP(hipMemcpyAsync(world4.data(), c->d_world4, sizeof(float4) * N, hipMemcpyDeviceToHost, c->stream));
  } else {
    MALIO_HIP(hipMemcpyAsync(world.data(), c->d_world, sizeof(float) * 3 * N, hipMemcpyDeviceToHost, c->stream));
  }
  MALIO_HIP(hipMemcpyAsync(plane.data(), c->d_plane, sizeof(float4) * N, hipMemcpyDeviceToHost, c->stream));
  std::vector<float4> near;
  if ((nearest || nearest_count) && c->nbr_epoch != c->map_epoch) {
    c->err = "malio_scan_get: the map changed after the last search pass; read Nearest_Points before map_add/delete";
    return MALIO_ERR_BAD_ARG;
  }
  std::vector<int> near_cnt;
  if (nearest || nearest_count) {
    if (int rcs = map_sync_search(c)) return rcs;
    if (c->nbr_epoch != c->map_epoch) {  // (a rebuild renumbered the map)
      c->err = "malio_scan_get: the map changed after the last search pass; read Nearest_Points before map_add/delete";
      return MALIO_ERR_BAD_ARG;
    }
    ArenaScope sc(c->arena);
    float4 *d_near = nullptr;
    u32 *d_far = nullptr;
    int *d_cnt = nullptr;
    MALIO_HIP(sc.get(&d_near, 5 * (size_t)N));
    MALIO_HIP(sc.get(&d_far, 5 * (size_t)N));
    MALIO_HIP(sc.get(&d_cnt, (size_t)N));
    if (int rcf = far_knn5(c, d_far)) return rcf;
    hipLaunchKernelGGL(k_gather_side, dim3((N + BLK - 1) / BLK), dim3(BLK), 0, c->stream, N, c->d_perm, c->d_nbr, d_far,
                       c->d_nfound, c->d_map_in, d_near, d_cnt);
    near.resize((size_t)5 * N), near_cnt.resize(N);
    MALIO_HIP(hipMemcpyAsync(near.data(), d_near, sizeof(float4) * near.size(), hipMemcpyDeviceToHost, c->stream));
    MALIO_HIP(hipMemcpyAsync(near_cnt.data(), d_cnt, sizeof(int) * N, hipMemcpyDeviceToHost, c->stream));
    MALIO_HIP(hipStreamSynchronize(c->stream));
  }
  MALIO_HIP(hipStreamSynchronize(c->stream));
  for (int i = 0; i < N; i++) {
    const u32 o = perm[i];
    if (normal_y) {
      // laserMapping.cpp:699,730,741: the last pass' rewrite is still pending (see commit_normal_y);
      // untouched when that pass bailed out (:635-639) or, for accepted points, extrinsic_est_en is off
      bool untouched = (c->last_M <= 0) || (sel[i] && !c->prm.extrinsic_est_en);
      normal_y[o] = untouched ? ny[i] : (float)tr[i];
    }
    if (nearest_count) nearest_count[o] = near_cnt[o];
    if (selected) selected[o] = sel[i];
    if (res_last) res_last[o] = sel[i] ? fabsf(pd2[i]) : 0.f;
    if (world_xyz) {
      if (from_search)
        world_xyz[3 * o] = world4[i].x, world_xyz[3 * o + 1] = world4[i].y, world_xyz[3 * o + 2] = world4[i].z;
      else
        world_xyz[3 * o] = world[i], world_xyz[3 * o + 1] = world[N + i], world_xyz[3 * o + 2] = world[2 * N + i];
    }
    if (normvec4) {
      normvec4[4 * o] = plane[i].x, normvec4[4 * o + 1] = plane[i].y, normvec4[4 * o + 2] = plane[i].z;
      normvec4[4 * o + 3] = pd2[i];
    }
  }
  if (nearest) {
    for (size_t k = 0; k < (size_t)5 * N; k++) {
      malio_point_t p;
      memset(&p, 0, sizeof(p));
      p.x = near[k].x, p.y = near[k].y, p.z = near[k].z, p._pad0 = 1.f, p.normal_y = near[k].w;
      nearest[k] = p;
    }
  }
  return MALIO_OK;
}

int malio_update_iterated(malio_handle_t h, malio_state_t *x, double *P, double R, int *stats, double *solve_time) {
  if (check(h) || !x || !P) return MALIO_ERR_BAD_ARG;
  MALIO_HIP_H(hipSetDevice(h->device));
  // The whole loop as one enqueued chain with the filter algebra on the device (csrc/ieskf_dev.hip), unless something
  // needs the host between passes: a pass hook, profiling of single passes, or - found out by the first valid pass -
  // fewer accepted points than states (the M x M form of esekfom.hpp:574-582 works on rows).
  if (h->update_mode != MALIO_UPDATE_HOST && !h->pass_hook && !h->profiling) {
    if (solve_time) *solve_time = 0;
    const int rc = h->update_mode == MALIO_UPDATE_GATED ? ieskf_update_gated(h, nullptr, x, P, stats, solve_time) : ieskf_update_device(h, x, P, stats);
    if (rc != MALIO_SMALL_M_FALLBACK) return rc;
  }
  return ieskf_update(h, nullptr, x, P, R, stats, solve_time);
}

// The update without the host in the loop, in two calls (csrc/ieskf_dev.hip: device-resident loop): `begin` enqueues all
// max_iteration + 1 passes with the n x n algebra of every iteration as kernels and returns; the caller's thread is free
// until it calls `end`, which waits and returns what malio_update_iterated returns. MALIO_SMALL_M_FALLBACK from `end`
// (a pass accepted fewer points than there are states): x and P are untouched, call malio_update_iterated.
int malio_update_iterated_begin(malio_handle_t h, const malio_state_t *x, const double *P) {
  if (check(h) || !x || !P) return MALIO_ERR_BAD_ARG;
  MALIO_HIP_H(hipSetDevice(h->device));
  if (h->dev_update_pending) {
    h->err = "malio_update_iterated_begin: the previous update has not been ended";
    return MALIO_ERR_BAD_ARG;
  }
  return ieskf_update_device_begin(h, x, P);
}
int malio_update_iterated_end(malio_handle_t h, malio_state_t *x, double *P, int *stats) {
  if (check(h) || !x || !P) return MALIO_ERR_BAD_ARG;
  MALIO_HIP_H(hipSetDevice(h->device));
  return ieskf_update_device_end(h, x, P, stats);
}

int malio_set_update_mode(malio_handle_t h, int mode) {
  if (check(h) || (mode != MALIO_UPDATE_DEVICE && mode != MALIO_UPDATE_HOST && mode != MALIO_UPDATE_GATED)) return MALIO_ERR_BAD_ARG;
  h->update_mode = mode;
  return MALIO_OK;
}

// Diagnostics (not part of the reference interface): {level-1 directory cells, map points, level-2 directory cells}.
int malio_debug_counters(malio_handle_t h, int *out8) {
  if (check(h) || !out8) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  if (int rc = map_apply_finish(c)) return rc;
  out8[0] = (int)c->nl1.ncells, out8[1] = c->map_n - c->map_dead, out8[2] = (int)c->nl2.ncells;
  out8[3] = c->n_rebuilds, out8[4] = c->n_inplace, out8[5] = c->map_dead, out8[6] = c->nl_tomb, out8[7] = c->map_n;
  return MALIO_OK;
}

// Diagnostics: {passes run as one kernel (k_pass), extrema guesses that held, guesses that missed (rows redone), gated
// updates that fell back to the host-driven loop because a gate timed out}.
int malio_debug_fuse_stats(malio_handle_t h, int *out4) {
  if (check(h) || !out4) return MALIO_ERR_BAD_ARG;
  out4[0] = h->fuse_passes, out4[1] = h->fuse_hits, out4[2] = h->fuse_misses, out4[3] = h->gate_timeouts;
  return MALIO_OK;
}

// Diagnostics: how the last search pass left the scan points - out8[k] = points with k neighbours inside sqrt(5) m
// (k = 0..5), [6] = unused, [7] = not served here (another shard's).
int malio_debug_nfound_hist(malio_handle_t h, int *out8) {
  if (check(h) || !out8) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  for (int k = 0; k < 8; k++) out8[k] = 0;
  if (c->N <= 0 || !c->scan_sorted) return MALIO_ERR_NO_SCAN;
  MALIO_HIP(hipSetDevice(c->device));
  std::vector<unsigned char> nf(c->N);
  MALIO_HIP(hipMemcpyAsync(nf.data(), c->d_nfound, c->N, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  for (unsigned char v : nf) out8[v <= 5 ? v : 7]++;
  return MALIO_OK;
}

// Diagnostics: what the last search pass did with the cached neighbours - {points, kept (no list walk), walked, the pass was
// allowed to keep at all}. Points of other shards count as neither.
int malio_debug_skip_stats(malio_handle_t h, int *out4) {
  if (check(h) || !out4) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  for (int k = 0; k < 4; k++) out4[k] = 0;
  if (c->N <= 0 || !c->scan_sorted || !c->d_kept) return MALIO_ERR_NO_SCAN;
  MALIO_HIP(hipSetDevice(c->device));
  std::vector<unsigned char> kept(c->N), nf(c->N);
  MALIO_HIP(hipMemcpyAsync(kept.data(), c->d_kept, c->N, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipMemcpyAsync(nf.data(), c->d_nfound, c->N, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  out4[0] = c->N, out4[3] = c->last_search_skip;
  for (int i = 0; i < c->N; i++) {  // (d_kept is written by the kernels of a handle with the option on only)
    if (c->last_search_skip && kept[i]) out4[1]++;
    else if (nf[i] <= 5) out4[2]++;
  }
  return MALIO_OK;
}

// Diagnostics: the level-1 neighbour lists as the next search would find them - {lists, lists flagged "in order of distance from
// the cell centre", flagged lists that are NOT, live entries}. out4[2] != 0 would be a bug (a search may end early in a flagged list).
int malio_debug_list_order(malio_handle_t h, long long *out4) {
  if (check(h) || !out4) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  for (int k = 0; k < 4; k++) out4[k] = 0;
  MALIO_HIP(hipSetDevice(c->device));
  if (int rc = map_sync_search(c)) return rc;  // (lists that a change of the map left stale are built first; queued maintenance is waited for)
  if (!c->nl1.table) return MALIO_OK;
  return nl_check_order(c, c->nl1, out4);
}

int malio_set_pass_hook(malio_handle_t h, void (*fn)(int, void *), void *user) {
  if (check(h)) return MALIO_ERR_BAD_ARG;
  h->pass_hook = fn, h->pass_hook_user = user;
  return MALIO_OK;
}

int malio_ieskf_step(int lid_num, int max_iteration, double limit, int iter_index, malio_state_t *x, const malio_state_t *x_propagated,
                     const double *P_propagated, const double *HtRinvH, const double *HtRinvh, int *t_io,
                     int *converge_out, int *done_out, double *P_out) {
  if (lid_num < 1 || lid_num > MALIO_MAX_LIDAR || !x || !x_propagated || !P_propagated || !HtRinvH || !HtRinvh ||
      !t_io || !converge_out || !done_out || !P_out)
    return MALIO_ERR_BAD_ARG;
  return ieskf_step(lid_num, max_iteration, limit, iter_index, x, x_propagated, P_propagated, HtRinvH, HtRinvh, t_io,
                    converge_out, done_out, P_out);
}

double malio_localize_weight(const double NtN6[6], double thresh_min, double thresh_max, double cov_min, double cov_max) {
  if (!NtN6) return 0.0;
  return mf::localize_weight(NtN6[0], NtN6[1], NtN6[2], NtN6[3], NtN6[4], NtN6[5], thresh_min, thresh_max, cov_min, cov_max);
}

int malio_result_buffer(malio_handle_t h, double **host, double **dev, int *len_doubles) {
  if (check(h) || !host || !dev || !len_doubles) return MALIO_ERR_BAD_ARG;
  if (!h->h_res) return MALIO_ERR_NO_SCAN;
  *host = h->h_res, *dev = h->d_res, *len_doubles = sums_len(h) + 16;
  return MALIO_OK;
}

int malio_scan_order(malio_handle_t h, int mode) {
  if (check(h) || mode < MALIO_SCAN_ORDER_AUTO || mode > MALIO_SCAN_ORDER_KEEP) return MALIO_ERR_BAD_ARG;
  h->scan_order_mode = mode;
  return MALIO_OK;
}

int malio_host_alloc(size_t bytes, void **out) {
  if (!out || bytes == 0) return MALIO_ERR_BAD_ARG;
  *out = nullptr;
  return hipHostMalloc(out, bytes, hipHostMallocPortable) == hipSuccess ? MALIO_OK : MALIO_ERR_ALLOC;
}
int malio_host_free(void *p) {
  if (!p) return MALIO_OK;
  return hipHostFree(p) == hipSuccess ? MALIO_OK : MALIO_ERR_HIP;
}

int malio_predict(int lid_num, malio_state_t *x, double *P, double dt, const double *Q, const double *acc,
                  const double *gyro) {
  if (lid_num < 1 || lid_num > MALIO_MAX_LIDAR || !x || !acc || !gyro || (P && !Q)) return MALIO_ERR_BAD_ARG;
  return predict_step(lid_num, x, P, dt, Q, acc, gyro);
}

}  // extern "C"
