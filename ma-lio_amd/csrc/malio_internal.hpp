// libmalio_hip internal declarations (gfx950 only). Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "../../include/malio.h"

namespace malio {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr u64 EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;
constexpr int BLK = 256;  // 4 wave64 per workgroup

// One spatial-hash entry: 16 B, read with a single dwordx4 load.
struct alignas(16) Cell {
  u64 key;
  u32 start;
  u32 count;
};

// Result of grouping a point set by hash cell (CSR over a compact open-addressing table).
struct CellGrid {
  Cell *table = nullptr;  // [tsize]
  u32 tmask = 0;          // tsize - 1
  u32 ncells = 0;
  float4 *pts = nullptr;  // [n] sorted by cell: x, y, z, bits(original index)
  u32 *orig = nullptr;    // [n] original index of each sorted point
  int n = 0;
  size_t cap_pts = 0, cap_table = 0;
};

// Neighbour lists (fast path of the 5-NN): for every fine cell (edge cf = cell / 1.5) that has a map point in
// its 3x3x3 block, the points of that block stored CONTIGUOUSLY (each map point is replicated 27 times -
// 432 MB at 1M points: this is what the 288 GB of HBM are for). A query then needs ONE directory probe and one
// streaming read of ~20 points instead of 27 probes and 9 scattered runs.
// Level-1 lists are kept SORTED by the distance of their entries from the centre of the list's cell (round 5, map_hash.hip:
// k_nl_sort): a query that has seen the first 32 entries knows a lower bound on the distance of everything behind them and
// mostly need not read on (measure.hip: nl_walk, EARLY). The top bit of Cell::count says "this list is sorted"; every reader of a
// neighbour list's count masks it. (The voxel grids' Cells never carry it.)
constexpr u32 NL_SORTED = 0x80000000u, NL_COUNT = 0x7FFFFFFFu;
constexpr u32 NL_SORT_MAX = 256;    // longer lists stay unsorted (and are walked whole): one wave sorts a list in registers
constexpr size_t NL_GUARD = 1024;  // entries allocated behind NList::pts[cap_pts): a walk's last round may read (never use) them
struct NList {
  Cell *table = nullptr;  // fine cell -> (start, count) of its neighbourhood list
  u32 tmask = 0;
  u32 ncells = 0;
  float4 *pts = nullptr;  // [cap_pts] x, y, z, bits(map index); a deleted point's entries carry x = +inf
  u32 *cap = nullptr;     // [table size] capacity of every list (count + slack): room for incremental inserts
  u32 *inc = nullptr;     // [table size] entries the batch being applied brings to each list (zero between batches)
  u32 *state = nullptr;   // device: [0] bump cursor into the tail of pts, [1] overflow flag, [2] cells, [3] lists on `work`
  u32 *work = nullptr;    // [work_cap][8] (sorted level only) the lists the batch being applied appends to: slot, count word before the
                          // batch, count after it, start | cell key (2 words), - , -  (k_nl_place -> k_nl_sort); grown by nl_ensure
  size_t work_cap = 0;
  size_t total = 0;       // entries reserved by the lists built last (capacities)
  size_t entries = 0;     // live entries at build time (27 per point for whole blocks; ~20.6 when pruned)
  size_t cap_pts = 0, cap_table = 0;
  float cf = 0.75f, inv_cf = 1.f / 0.75f;
  bool pruned = false;    // lists hold the points within one cell edge of the cell instead of the whole 3x3x3 block
  bool sorted = false;    // lists are kept sorted by distance from the cell centre (NL_SORTED)
};

// scratch of build_nlist (open-addressing directory under construction), kept between rebuilds
struct NlScratch {
  u64 *keys = nullptr;
  u32 *cnt = nullptr, *start = nullptr, *capv = nullptr, *tiles = nullptr, *counters = nullptr;
  u32 cap = 0;
};

// device-side view of a neighbour-list level for the incremental kernels
struct NlDev {
  Cell *table;
  u32 tmask;
  float4 *pts;
  u32 *cap, *inc, *state, *work;
  u32 bump_end, work_cap;
  float inv_cf, cf;
  int sorted;
  int pruned;  // level 1: a list holds only the block's points within one cell edge of its cell (nl_member)
};

// NL_REACH: how far from a cell (in cell edges) a point may lie and still be in the cell's PRUNED list; the certificate of a
// search in that list is NL_REACH x cf + (distance to the nearest face). 1: every point within one cell edge (rounds 2-4).
#ifndef NL_REACH
#define NL_REACH 1.0f
#endif
#if defined(__HIP__)
// Which of the 27 lists around a point's own cell hold it. Unpruned: all of them (the list of a cell is the whole 3x3x3
// block around it). Pruned (level 1): only the cells the point is within one cell edge of. A search in cell c certifies
// its result against g1 = cf + (distance of the query to the nearest face of c) - margins; a point p within that radius of
// a query q in c satisfies dist(p, c) <= |p - q| - (way out of c along the segment) <= cf, so nothing a certificate
// relies on is dropped - and for positions uniform inside their cells only 1 + 6 + 12 (pi/4) + 8 (pi/6) = 20.6 of the 27
// neighbours qualify: lists, their traffic and the candidates per query shrink by 24 %. The tolerance is twice the
// search's allowance for the float rounding of cell coordinates; every kernel that adds, finds or removes an entry
// evaluates this same function on the same stored coordinates.
__device__ __forceinline__ bool nl_member(int pruned, float gx, float gy, float gz, int ix, int iy, int iz, int dx, int dy,
                                          int dz) {
  if (!pruned) return true;
  const float fx = gx - (float)ix, fy = gy - (float)iy, fz = gz - (float)iz;
  const float ax = dx == 0 ? 0.f : (dx > 0 ? 1.f - fx : fx), ay = dy == 0 ? 0.f : (dy > 0 ? 1.f - fy : fy),
              az = dz == 0 ? 0.f : (dz > 0 ? 1.f - fz : fz);
  const float reach = NL_REACH + 1e-5f + 6e-7f * (fabsf(gx) + fabsf(gy) + fabsf(gz) + 3.0f);
  return ax * ax + ay * ay + az * az <= reach * reach;
}
// cell directory hashing shared by every .hip file (measure.hip keeps identical _d copies next to its hot loops)
__device__ __forceinline__ u64 cell_key(int ix, int iy, int iz) {
  const u64 B = 1ull << 20;
  return ((u64)(ix + (long long)B) & 0x1FFFFF) | (((u64)(iy + (long long)B) & 0x1FFFFF) << 21) |
         (((u64)(iz + (long long)B) & 0x1FFFFF) << 42);
}
__device__ __forceinline__ u32 hash_key(u64 k) {  // 32-bit multiplicative mix (7 VALU ops); == hash_key_d()
  u32 lo = (u32)k, hi = (u32)(k >> 32);
  u32 h = lo * 0x9E3779B1u ^ hi * 0x85EBCA77u;
  h ^= h >> 15;
  h *= 0xC2B2AE3Du;
  h ^= h >> 13;
  return h;
}
#endif

// Spatial sharding of the map across `world` handles (BASELINE config 4): space is cut into tiles - cubes of
// edge tile_m, or (PartView::columns, the shape a ground vehicle's map wants: round 5) vertical columns over tile_m squares,
// which have no neighbours above or below and therefore no halo there: 8 shards of BASELINE config 4 store 1.8 x the map
// instead of 2.6 x (profiles/round5/r05_tile_shards.txt) -, a tile belongs to the shard its hashed coordinates name. A shard stores the map points of its own tiles plus a halo
// (every point whose voxel box, grown by PART_HALO, touches an owned tile) and serves the scan points whose world point
// of the SEARCH pass lies in an owned tile. PART_HALO > sqrt(5) m: every neighbour the reference can accept
// (pointSearchSqDis[4] <= 5, laserMapping.cpp:587) of an owned query is in the shard, so its result equals the one an
// unsharded engine gives; nothing needs merging. Same float arithmetic on host and device (-ffp-contract=off).
constexpr float PART_HALO = 2.3f;
struct PartView {
  int rank = 0, world = 1;  // world <= 1: not partitioned
  float inv_tile = 1.f / 16.f;
  int columns = 0;  // MALIO_TILE_COLUMNS: a tile is the whole vertical column over its (x, y) square - no halo above or below it
  int lat_k = 5;    // columns: owner = (tx + lat_k ty) mod world (part_lattice_k)
};
__host__ __device__ inline int tile_coord(float x, float inv_tile) { return (int)floorf(x * inv_tile); }
__host__ __device__ inline u32 tile_hash(int tx, int ty, int tz) {
  u32 h = (u32)tx * 0x9E3779B1u ^ (u32)ty * 0x85EBCA77u ^ (u32)tz * 0xC2B2AE3Du;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  return h;
}
__host__ __device__ inline u32 tile_owner(int tx, int ty, int tz, u32 world) { return tile_hash(tx, ty, tz) % world; }
// vertical tile coordinate under the partition's shape (columns: every height is tile 0)
__host__ __device__ inline int part_tz(const PartView &p, float z) { return p.columns ? 0 : tile_coord(z, p.inv_tile); }
// Who owns tile (tx, ty, tz)? Cubes: the hash above. Columns: the lattice (tx + k ty) mod world - a scan covers ~150 columns
// of 16 m, too few for a hash to balance over 8 shards (max / mean points served 1.45); on the lattice every run of `world`
// columns along x holds one of each shard and the rows are shifted by k: 1.04 at 16 m, 1.05 at 24 m
// (profiles/round5/r05_tile_shards.txt). k: the smallest of 5, 3, 7, 11, 13 that shares no factor with `world`.
inline int part_lattice_k(int world) {
  const int cand[5] = {5, 3, 7, 11, 13};
  for (int c = 0; c < 5; c++) {
    int a = cand[c], b = world;
    while (b) { const int t = a % b; a = b, b = t; }
    if (a == 1) return cand[c];
  }
  return 1;
}
__host__ __device__ inline u32 part_tile_owner(const PartView &p, int tx, int ty, int tz) {
  if (!p.columns) return tile_owner(tx, ty, tz, (u32)p.world);
  const long long v = ((long long)tx + (long long)p.lat_k * (long long)ty) % (long long)p.world;
  return (u32)(v < 0 ? v + p.world : v);
}
__host__ __device__ inline u32 part_owner_of(const PartView &p, float x, float y, float z) {
  return part_tile_owner(p, tile_coord(x, p.inv_tile), tile_coord(y, p.inv_tile), part_tz(p, z));
}
__host__ __device__ inline bool part_owns(const PartView &p, float x, float y, float z) {
  return p.world <= 1 || part_owner_of(p, x, y, z) == (u32)p.rank;
}
// does the box [c - r, c + r] touch a tile of this shard? (at most 8 tiles while 2 r < tile edge)
__host__ __device__ inline bool part_touches(const PartView &p, float cx, float cy, float cz, float r) {
  if (p.world <= 1) return true;
  const int x0 = tile_coord(cx - r, p.inv_tile), x1 = tile_coord(cx + r, p.inv_tile);
  const int y0 = tile_coord(cy - r, p.inv_tile), y1 = tile_coord(cy + r, p.inv_tile);
  const int z0 = part_tz(p, cz - r), z1 = part_tz(p, cz + r);
  for (int z = z0; z <= z1; z++)
    for (int y = y0; y <= y1; y++)
      for (int x = x0; x <= x1; x++)
        if (part_tile_owner(p, x, y, z) == (u32)p.rank) return true;
  return false;
}
// a map point is stored by every shard its down-sampling voxel (edge fs; the point itself when fs <= 0) reaches: all
// points of one voxel live on the same shards, so the per-voxel keeper rule of Add_Points sees complete voxels everywhere
__host__ __device__ inline bool part_stores(const PartView &p, float x, float y, float z, float fs) {
  if (p.world <= 1) return true;
  if (!(fs > 0.f)) return part_touches(p, x, y, z, PART_HALO);
  const float hx = (floorf(x / fs) + 0.5f) * fs, hy = (floorf(y / fs) + 0.5f) * fs, hz = (floorf(z / fs) + 0.5f) * fs;
  return part_touches(p, hx, hy, hz, PART_HALO + 0.5f * fs);
}

// Per-LiDAR constants of one pass (all double; rotation matrices row-major).
struct LidarConst {
  double Rl[9], tl[3];    // extrinsic of this LiDAR (iterated)   q_l, t_l
  double Rtc[9], ttc[3];  // temporal compensation (identity for lid 0)
};
struct PassConst {
  double Rw[9], pw[3];  // s.rot, s.pos
  double R0[9], t0[3];  // extrinsic 0
  LidarConst lid[MALIO_MAX_LIDAR];
  int L, extrinsic_est_en;
  float plane_th;
  double cov_threshold;
};
struct WeightConst {
  double plane_cov_max, plane_cov_min, point_cov_max, point_cov_min, range_min, range_max;
};

// Quaternion form of the state of one pass (stage 1: world transform in Eigen's quaternion * vector order)
struct D3 {
  double x, y, z;
};
struct Q4 {
  double x, y, z, w;
};
struct QuatConst {
  Q4 rot;
  D3 pos;
  Q4 q0;
  D3 t0;
  Q4 ql[MALIO_MAX_LIDAR];
  D3 tl[MALIO_MAX_LIDAR];
  Q4 qtc[MALIO_MAX_LIDAR];  // index lid (0 unused)
  D3 ttc[MALIO_MAX_LIDAR];
};

// Device-resident state of the iterated update (csrc/ieskf_dev.hip): with it a whole update_iterated is ONE chain of
// kernels the host enqueues up front and waits for once. The pass kernels (DEV = true instantiations) read the state
// and the control words from here instead of from their kernel arguments; k_ieskf_step - one workgroup running the
// n x n filter algebra of esekfom.hpp:521-720 after every pass - is the only writer.
constexpr int DEV_NMAX = 17 + 6 * MALIO_MAX_LIDAR;  // 41
struct DevLoop {
  int done;         // the loop is over (or was never started): every remaining kernel of the chain exits at once
  int converge;     // ekfom_data.converge of the NEXT pass: 1 = search, 0 = reuse (esekfom.hpp:649-663)
  int i;            // loop index of the NEXT pass (esekfom.hpp:509: -1 .. maximum_iter - 1)
  int t;            // converged iterations so far (:658)
  int passes, searches, lastM;
  int status;       // MALIO_OK, MALIO_SMALL_M_FALLBACK (n > M: the host redoes the update on the rows path) or < 0
  int mm_parity;    // extrema slot set the NEXT pass accumulates into (the other one is cleared by it)
  int commit_prev;  // the previous pass was valid: its (sel, trace) are folded into normal_y by the next one
  int valid_any;    // a pass was valid: P_proj of the last valid iteration is what a loop that runs out leaves behind
  int lastM_valid;  // accepted points of the last VALID pass (stats[2])
  int last_search;  // the last pass that ran was a search pass (feats_down_world then lives in world4)
  int maximum_iter, L, extrinsic_est_en;
  int search_skip;  // the NEXT pass, if a search pass, may keep cached neighbours (search_wg phase A')
  int skip_opt;     // device-resident loop: what search_skip becomes once the loop's first search pass has run
  double limit;
  double mm_guess[4];  // one-kernel pass (k_pass): the extrema the rows are weighted with, [max_u, -min_u, max_R, -min_R]
  double tcq[MALIO_MAX_LIDAR][4], tct[MALIO_MAX_LIDAR][3];  // temporal compensation of this scan (index lid - 1)
  malio_state_t x, x_prop;
  QuatConst qc;  // x in the forms the pass kernels read
  PassConst pc;
  // localization-weight parameters (laserMapping.cpp:749-756)
  double loc_thresh_min, loc_thresh_max, loc_cov_min, loc_cov_max;
  long long stamps[16];  // developer aid: wall_clock64 (100 MHz) at the phase boundaries of the last step kernel
};

// Uncertainty-table entry folded for trace(Sigma_p) (associate_uct.hpp:153-175):
// p' = T (0.05 p, 1);  trace = k0 + lin . p' + p'^T Q p'
struct alignas(16) UncEntry {
  double T[12];  // top 3 rows of pose.T_
  double k0;
  double lin[3];
  double Q[6];  // xx yy zz xy xz yz
};

// Gate between two passes of the gated update loop (csrc/ieskf_dev.hip): announce the finished pass to the host, wait for
// the control block of the next one, copy it into the DevLoop the pass kernels read.
struct GateArgs {
  DevLoop *dl = nullptr;
  const double *cmd = nullptr;   // pinned, or device memory the host stores into: the control block, DevLoop layout (rounded up to 256 B) ...
  const int *cmd_seq = nullptr;  // ... and the word the host stores LAST (release)
  int *msg_seq = nullptr;        // pinned: sequence word the GPU publishes
  u32 *ticket = nullptr;         // device counter of the kernel the gate rides on (k_final_reduce: its last workgroup is the gate)
  int publish = 0, wait_for = 0, ndoubles = 0;
  long long timeout_ticks = 0;  // 100 MHz ticks the gate waits for the host (0: GATE_TIMEOUT_US)
};
constexpr long long GATE_TIMEOUT_US = 200000;  // default; MALIO_GATE_TIMEOUT_MS overrides it per handle
#if defined(__HIP__)
// called by every thread of ONE workgroup (256 threads); a gate gives up after its timeout (the host died, returned, or
// was descheduled for that long): the rest of the chain then drains as on `done`, and a host that is still there redoes
// the update with the host-driven loop (ieskf_update_gated)
__device__ inline void gate_body(const GateArgs &g) {
  __shared__ int s_gate_ok;
  if (threadIdx.x == 0) {
    if (g.publish) {
      __threadfence_system();
      __hip_atomic_store(g.msg_seq, g.publish, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(g.cmd_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != g.wait_for) {
      if (wall_clock64() - t0 > (g.timeout_ticks > 0 ? g.timeout_ticks : GATE_TIMEOUT_US * 100)) {  // 100 MHz
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    s_gate_ok = ok;
    if (!ok) g.dl->done = 1, g.dl->status = MALIO_ERR_TIMEOUT;
  }
  __syncthreads();
  if (!s_gate_ok) return;
  double *dst = reinterpret_cast<double *>(g.dl);
  for (int e = threadIdx.x; e < g.ndoubles; e += blockDim.x) dst[e] = g.cmd[e];
}
#endif

// Per-call device scratch: one block that grows to the largest call seen, handed out by a bump pointer and released
// in stack order (hipMalloc/hipFree cost ~0.1 ms each and a map update needs a dozen temporaries).
struct Arena {
  char *base = nullptr;
  size_t cap = 0, off = 0, want = 0;
  int depth = 0;
  std::vector<void *> extra;  // overflow blocks of the current outermost scope
  hipError_t take_bytes(void **out, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    want += bytes;
    if (off + bytes <= cap) {
      *out = base + off;
      off += bytes;
      return hipSuccess;
    }
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e == hipSuccess) extra.push_back(q);
#ifdef MALIO_POISON
    // (hipMemset on device memory may return before it has run, on the NULL stream - which the handle's non-blocking streams do
    // not wait for: without the synchronisation the poison lands on top of what the next kernels write)
    if (e == hipSuccess) (void)hipMemset(q, 0xFF, bytes), (void)hipDeviceSynchronize();
#endif
    *out = q;
    return e;
  }
  void release_all() {
    for (void *q : extra) (void)hipFree(q);
    extra.clear();
    if (base) (void)hipFree(base);
    base = nullptr, cap = 0, off = 0;
  }
};
struct ArenaScope {  // allocations made through a scope die with it
  Arena &a;
  size_t mark, want_mark;
  explicit ArenaScope(Arena &ar) : a(ar), mark(ar.off), want_mark(ar.want) { a.depth++; }
  ~ArenaScope() {
#ifdef MALIO_POISON
    // developer build (make POISON=1): what a scope hands back reads as 0xFF.. from then on - a kernel that still uses a
    // temporary of a scope that has ended, or slack nobody wrote, computes with NaNs / indices of 4 G instead of with whatever
    // the last user left there. Blocking on purpose (the queued kernels of this scope finish first).
    if (a.base && a.off > mark) {
      (void)hipDeviceSynchronize();
      (void)hipMemset(a.base + mark, 0xFF, a.off - mark);
      (void)hipDeviceSynchronize();
    }
#endif
    a.off = mark;
    if (--a.depth == 0) {
      if (!a.extra.empty()) {  // the block was too small: grow it for the next call
        const size_t need = a.want + a.want / 2;
        a.release_all();
        if (hipMalloc((void **)&a.base, need) == hipSuccess) a.cap = need;
#ifdef MALIO_POISON
        if (a.cap) (void)hipMemset(a.base, 0xFF, a.cap), (void)hipDeviceSynchronize();
#endif
      }
      a.want = 0;
    } else {
      (void)want_mark;  // inner scopes keep counting towards the outermost call's demand
    }
  }
  template <class T>
  hipError_t get(T **out, size_t count) {
    return a.take_bytes((void **)out, sizeof(T) * (count ? count : 1));
  }
};

// One scan point as it crosses PCIe (20 B instead of the 48 B of pcl::PointXYZINormal), in the caller's order:
// body-frame xyz, the packed (LiDAR slot | int(normal_x) << 8) word of laserMapping.cpp:570,694,737 and the input
// normal_y (returned untouched where the reference does not write it). Grouping by LiDAR is left to the scan sort.
struct UploadRec {
  float x, y, z;
  u32 w;
  float ny;
};

// undistorted cloud of one LiDAR kept in HBM between malio_undistort_resident and malio_scan_set_resident
struct ResCloud {
  float *d = nullptr;  // [n][12] pcl::PointXYZINormal layout
  size_t cap = 0;
  int n = 0;
};

struct Ctx {
  malio_params_t prm{};
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::string err;
  float cell = 1.125f, inv_cell = 1.f / 1.125f;
  PartView part;  // malio_set_partition
  bool part_sentinel = false;  // this shard's part of the map was empty: it holds one far-away placeholder point

  // map
  NlScratch nl_scratch;
  Arena arena;       // per-call temporaries of the map update paths
  ResCloud res[MALIO_MAX_LIDAR];
  void *h_stage = nullptr;  // pinned upload staging (map_build, scan_set), grown on demand
  size_t cap_stage = 0;
  // malio_measure_node: extrema the rows are weighted with (device copy + pinned staging), the guess carried from pass
  // to pass of one scan
  double *d_node_mm = nullptr, *h_node_mm = nullptr;
  double node_guess[4] = {0, 0, 0, 0}, node_uploaded[4] = {0, 0, 0, 0};
  bool node_guess_valid = false, node_uploaded_valid = false;
  int node_hits = 0, node_misses = 0;
  double node_mode_word = 0.0;  // which loop this shard's update runs (1 gated chain, 2 pass by pass; 0: a bare malio_measure_node): word 5 of its pass-0 row
  u32 *h_mbox = nullptr, *d_mbox = nullptr;  // pinned + device alias: small results for the host (counts), written
                                             // by kernels or copies; one stream sync serves all
  bool stage_pending = false;  // an async copy out of h_stage may still be in flight on `stream`
  hipEvent_t ev_upload = nullptr;  // recorded behind the DMA copy of a page-locked cloud (malio_scan_set): until then the
  bool upload_in_flight = false;   // caller's buffer is in use (malio_scan_upload_wait)
  int scan_set_sync = 0;           // MALIO_OPT_SCAN_SET_SYNC: malio_scan_set waits for that copy itself
  CellGrid gnew;     // new points of an Add_Points call grouped by downsample voxel (buffers reused)
  bool apply_pending = false;  // an in-place list update is queued, its verdict (fitted / overflowed) not read yet
  // map_apply's kernels (tombstones, kill, append, list maintenance) run on a stream of their own: the next scan's upload,
  // pack and grouping do not read the map and overlap with them; whoever reads or writes the map or the lists on `stream`
  // joins first (maint_join: map_sync_search and every map entry point). Their inputs live in arena_maint until the
  // next mutator has seen ev_maint_done complete.
  hipStream_t maint_stream = nullptr;
  hipEvent_t ev_maint_in = nullptr, ev_maint_done = nullptr;
  bool maint_pending = false;   // `stream` has not waited for ev_maint_done yet
  bool maint_inflight = false;  // the host has not seen ev_maint_done complete yet (arena_maint in use)
  int maint_enabled = 1;        // MALIO_OPT_MAINT_STREAM = 0: everything on `stream` (A/B)
  Arena arena_maint;
  ArenaScope *maint_scope = nullptr;
  Cell *d_small_table = nullptr;  // map_incremental's usual batch: voxel table + member lists of k_group_small
  u32 *d_small_orig = nullptr;
  int mapinc_small = 4096;  // MALIO_OPT_MAPINC_SMALL: cap of the one-read-back path (SMALL_CAP; 0: general path always)
  u32 small_seq = 0;
  bool search_dirty = false;  // d_map_in changed; nl1/nl2 are rebuilt by the next search (map_sync_search)
  NList nl1, nl2;  // level 1: cf1 = cell_size (fast path); level 2: cf2 = 2 * cell_size >= sqrt(5) (always exact)
  float4 *d_map_in = nullptr;  // [map_n] map array: x y z normal_y, slot index = map id (plane fit + Nearest_Points)
  float4 *d_map_alt = nullptr;  // compaction target of map_add / map_delete_boxes (swapped with d_map_in)
  size_t cap_map_in = 0, cap_map_alt = 0;
  // The map array in CELL ORDER (round 6, MALIO_OPT_MAP_CELL_ORDER): a rebuild leaves slots [0, map_sorted_n) sorted by their
  // level-1 cell, columns of cells in the scan grouping's order (map_update.hip: map_rebuild_search), so that the five neighbours a query gathers for its plane fit
  // share one or two 128-byte lines; d_map_ord[slot] is the slot's rank in INSERTION order (what "lowest map index" means to the
  // keeper rule's ties, k_vox_add, and the order malio_map_get hands the map out in). Slots appended since are their own rank.
  u32 *d_map_ord = nullptr;
  size_t cap_map_ord = 0;
  int map_sorted_n = 0;
  unsigned char *d_del = nullptr;  // per-slot "deleted by this batch" marks of the voxel update; all zero between calls (k_map_kill_list clears what it kills)
  size_t cap_del = 0;
  int map_dead = 0;  // slots of d_map_in[0, map_n) that hold a deleted point (x = +inf)
  int nl_tomb = 0;   // tombstoned points in the lists since the last full build
  int n_rebuilds = 0, n_inplace = 0;  // diagnostics: full list builds / map changes applied to the lists in place
  int map_epoch = 0;   // bumped by every change of the map array (indices in d_nbr refer to one epoch)
  int nbr_epoch = 0;   // epoch d_nbr was filled in
  int map_n = 0;  // slots in use (live + dead); the number of valid points is map_n - map_dead
  // scan (device arrays in SORTED order: grouped by lidar, then by hash cell of the world position)
  int N = 0;
  bool scan_sorted = false;
  bool scan_keep_order = false;  // this scan is used in its upload order (no spatial sort), see malio_scan_order
  int scan_order_mode = 0;       // MALIO_SCAN_ORDER_*
  UploadRec *d_upload = nullptr;  // [N] scan as uploaded, caller's order (see UploadRec)
  float *d_raw = nullptr;         // [N][12] the caller's page-locked cloud as copied (malio_scan_set, pinned path)
  size_t cap_raw = 0;
  // malio_scan_stage: the NEXT scan's cloud copied ahead on a stream of its own (under map_incremental of the current scan)
  bool packinfo_clean = false;  // d_packinfo is all zero (left so by k_sort_scan)
  bool count_in_sort = false;  // malio_scan_set_packed (sorted scans): the per-slot counts are formed by k_sort_count
  void *d_ahead = nullptr;
  size_t cap_ahead = 0;            // bytes
  const void *ahead_ptr = nullptr;  // the caller's buffer the staged bytes came from (nullptr: nothing staged)
  int ahead_n = 0, ahead_packed = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_ahead = nullptr, ev_ahead_free = nullptr;
  bool ahead_busy = false;          // a consumer of d_ahead was enqueued on `stream` after ev_ahead_free was last recorded
  u32 *d_packinfo = nullptr;      // k_pack_raw: per-slot counts, bad slots, descents
  u32 *h_packinfo = nullptr, *d_packinfo_pub = nullptr;  // pinned copy (+ sequence word [15]) of k_pack_raw's counts
  u32 pack_seq = 0, apply_seq = 0;
  bool pack_publish_pending = false;  // the counts of k_pack_raw still have to be stored to pinned memory by a later kernel
  u32 *d_sort_cnt = nullptr;      // scan grouping: bucket counts + offsets (measure.hip sort_scan)
  bool seg_pending = false;       // seg_start[] not known yet: the counts are still on the device
  float4 *d_scan = nullptr;     // [N] sorted
  u32 *d_perm = nullptr;        // [N] sorted -> original index
  int last_M = -1;
  int seg_start[MALIO_MAX_LIDAR + 1] = {0};
  size_t cap_scan = 0;
  UncEntry *d_unc = nullptr;  // all tables concatenated
  int unc_off[MALIO_MAX_LIDAR] = {0}, unc_len[MALIO_MAX_LIDAR] = {0};
  size_t cap_unc = 0;
  double tcq[MALIO_MAX_LIDAR][4], tct[MALIO_MAX_LIDAR][3];  // temporal_comp (index lid-1)
  PassConst pc;  // matrix form of the state of the current pass (filled by stage 1)
  // per-point pass state (sorted order)
  u32 *d_nbr = nullptr;        // [5][N] map sorted index or 0xFFFFFFFF
  float4 *d_plane = nullptr;   // [N] pabcd
  float *d_pd2 = nullptr;      // [N]
  float *d_world = nullptr;    // [3][N]
  float4 *d_world4 = nullptr;  // [N] world point of the search pass (phase A of the search; k_far_nearest)
  float *d_ny = nullptr;       // [N] normal_y state (see commit_normal_y)
  uint4 *d_pcache = nullptr;   // [N] probe cache: cell key, start, count of the point's last level-1 directory probe (measure.hip, phase B)
  bool probe_valid = false;    // a search pass of this scan has filled d_pcache (search_skip_begin)
  float4 *d_cert = nullptr;    // [N] search-skip certificate (world point of the last list walk, radius free of outsiders)
  unsigned char *d_kept = nullptr;  // [N] the last search pass kept the point's cached neighbours
  bool cert_valid = false;     // a search pass of this scan has been queued: its certificates exist in stream order
  int last_search_skip = 0;    // the last search pass was allowed to keep cached neighbours
  double *d_ucov = nullptr;    // [N] unit_cov
  double *d_trace = nullptr;   // [N] trace(Sigma_p) (clamp rule by selected flag)
  unsigned char *d_sel = nullptr;  // [N]
  unsigned char *d_nfound = nullptr;  // [N]
  // reductions
  bool last_pass_search = false;
  u64 *d_mmslots = nullptr;  // [2 parities][64 slots][5]: max_u, min_u, max_R, min_R (order-encoded doubles), count
  int mm_parity = 0;
  // speculating pass (measure.hip, k_pass): the per-workgroup tiles, [NSUM][cap_tiles]
  double *d_tiles = nullptr;
  size_t cap_tiles = 0;
  double mm_guess[4] = {0, 0, 0, 0};  // true extrema of the last completed pass of this scan: the next pass' guess
  bool mm_guess_valid = false;
  double fuse_guess_used[4] = {0, 0, 0, 0};  // what the last k_pass launch was given
  int fuse_enabled = 1;  // MALIO_OPT_FUSE = 0 switches the one-kernel pass off
  bool fuse_debug_bad_guess = false;  // MALIO_OPT_DEBUG_FUSE_BAD_GUESS
  // options without a home above (malio_set_option; initial values from the environment, read once by malio_create)
  int opt_search_skip = 0;     // MALIO_OPT_SEARCH_SKIP (off: measured at +1 us per search pass for the few points it keeps)
  int opt_gate_pinned = 0;     // MALIO_OPT_GATE_PINNED
  int opt_nl_full_blocks = 0;  // MALIO_OPT_NL_FULL_BLOCKS
  int opt_nl_sorted = 1;       // MALIO_OPT_NL_SORTED
  int opt_probe_cache = 1;     // MALIO_OPT_PROBE_CACHE
  int opt_map_cell_order = 1;  // MALIO_OPT_MAP_CELL_ORDER (takes effect at the next rebuild)
  int opt_early_min_queries = 32768;  // MALIO_OPT_EARLY_MIN_QUERIES: scans of at least this many queries end walks of ordered lists early (measure.hip, view_l1)
  int opt_node_gated = 1;      // MALIO_OPT_NODE_GATED: a shard's update runs the gated chain (host exchanges only)
  int node_gated_runs = 0;     // updates of a shard that went through the gated chain
  int node_gated_redone = 0;   // updates the gated chain of a shard handed back to the per-pass loop
  int fuse_cooldown = 0;  // eligible passes left that do NOT speculate (set by a miss, see fuse_eligible)
  int fuse_cooldown_len = 3, fuse_hits_in_row = 0;  // (FUSE_COOLDOWN_MIN; adapted by fused_collect)
  int fuse_hits = 0, fuse_misses = 0, fuse_passes = 0;
  double *d_partials = nullptr;  // [NSUM][cap_partials]
  double *d_sums = nullptr;      // [NSUM_OUT]
  double *h_sums = nullptr;      // pinned
  double *h_res = nullptr, *d_res = nullptr;  // pinned + its device-visible alias: results stored by the kernels
  double *h_minmax = nullptr;    // pinned [5]
  // device-resident iterated update (csrc/ieskf_dev.hip)
  DevLoop *d_loop = nullptr;     // control block + state
  double *d_loopbuf = nullptr;   // P_prop | P_proj | P_projinv | dx_new | scratch (see ieskf_dev.hip)
  char *h_loop_in = nullptr;     // pinned: what one update uploads (DevLoop + P_prop)
  char *h_loop_out = nullptr;    // pinned, device-mapped: what the last step kernel stores (DevLoop + P)
  char *d_loop_out = nullptr;    // ... its device alias
  u32 *d_gate_ticket = nullptr;  // (own allocation, zeroed once)
  char *h_gate = nullptr, *d_gate = nullptr;  // pinned + device alias: control block and sequence words of the gated loop
  // Large BAR: the host stores the control block and its sequence word straight into (fine-grained) device memory, so the
  // gate polls and copies local memory instead of reading pinned host memory across PCIe (null: no large BAR, pinned path)
  char *d_cmd = nullptr;
  std::vector<double> gate_stage;  // the block as the host composes it, before it goes out in one piece
  int gate_epoch = 1;
  long long gate_timeout_ticks = 0;  // MALIO_OPT_GATE_TIMEOUT_MS in 100 MHz ticks (0: the default)
  int gate_debug_stall_ms = 0;       // MALIO_OPT_DEBUG_GATE_STALL_MS: the host sleeps before publishing pass 2 (tests the timeout path)
  bool d_cmd_tried = false;          // the fine-grained control block was asked for already (and refused, when d_cmd is null)
  int gate_timeouts = 0;             // updates that fell back to the host-driven loop because a gate gave up
  double gate_trace[60] = {0};  // developer aid: host-side timestamps of the last gated update
  int gate_trace_n = 0;
  int update_mode = MALIO_UPDATE_GATED;  // malio_set_update_mode
  bool dev_update_pending = false;       // malio_update_iterated_begin was called, _end not yet
  double *d_rows = nullptr;      // optional dense rows [N][C+2]
  size_t cap_rows = 0;
  size_t cap_partials = 0;
  void (*pass_hook)(int, void *) = nullptr;  // malio_set_pass_hook
  void *pass_hook_user = nullptr;
  // profiling
  bool profiling = false;
  std::vector<hipEvent_t> ev;
  std::vector<const char *> ev_names;
  int ev_used = 0;
  std::vector<float> last_ms;
  std::vector<const char *> last_names;
};

#define MALIO_HIP(call)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      c->err = std::string(#call) + ": " + hipGetErrorString(e_);                                \
      return MALIO_ERR_HIP;                                                                      \
    }                                                                                            \
  } while (0)

#define MALIO_HIP_H(call)                                                                        \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_);                                \
      return MALIO_ERR_HIP;                                                                      \
    }                                                                                            \
  } while (0)

// group `n` device points (float4, xyz used) by spatial-hash cell into `g` (allocates/grows g's buffers)
int group_by_cell(Ctx *c, const float4 *d_in, int n, float inv_cell, CellGrid &g, const u32 *d_in_orig = nullptr,
                  float div_cell = 0.f);  // div_cell > 0: cell index = floor(x / div_cell) (ikd-Tree voxel rule)
int exclusive_scan_u32(Ctx *c, const u32 *d_in, u32 *d_out, u32 *d_tiles, int n, u32 *total_out = nullptr,
                       const u32 *fwd = nullptr, int nfwd = 0);  // d_tiles: [(n+1023)/1024 + 1]
int exclusive_scan_u32_pair(Ctx *c, const u32 *inA, u32 *outA, u32 *tilesA, u32 *totalA, const u32 *inB, u32 *outB,
                            u32 *tilesB, u32 *totalB, int n);  // two scans of one length in one pair of launches
void free_grid(CellGrid &g);
int build_nlist(Ctx *c, const float4 *d_in, int n, float cf, NList &nl, bool pruned = false, bool sorted = false);
void free_nlist(NList &nl);
int nl_check_order(Ctx *c, NList &nl, long long out4[4]);  // diagnostics: lists, flagged sorted, flagged but out of order, live entries
// incremental maintenance of one level (kernels in map_hash.hip); overflow is reported through nl.state[1]
// the map array's share of a batch, done by a third slice of k_nl_ensure's grid: dlist[ndel] die, kept new points go to dst[rank]
struct MapSide {
  float4 *mapp;
  const u32 *dlist;
  int ndel;
  unsigned char *del;
  u32 del_n;
  const u32 *rank;
  float4 *dst;
};
void nl_ensure(Ctx *c, hipStream_t st, NList &nl_a, NList &nl_b, const float4 *d_new, const u32 *keep, int m,
               const MapSide &side);  // both levels at once
void nl_append(Ctx *c, hipStream_t st, NList &nl_a, NList &nl_b, const float4 *d_new, const u32 *keep, const u32 *rank, u32 og_base, int m);
void nl_tombstone(Ctx *c, hipStream_t st, NList &nl_a, NList &nl_b, const float4 *d_map, const u32 *dlist, int ndel);  // dlist: deleted map indices
int maint_join(Ctx *c);         // `stream` waits for the queued map maintenance (no host wait)
int maint_scope_begin(Ctx *c);  // a mutator's entry: recycle arena_maint once the previous batch is known to be done
void maint_destroy(Ctx *c);
void free_nl_scratch(NlScratch &s);

// capi.hip: the handle's pinned upload staging buffer (waits for an upload still in flight out of it)
int host_stage(Ctx *c, size_t bytes, void **out);
// map_update.hip
int map_add(Ctx *c, const float4 *h_pts, int n, int downsample_on, int *out_added);
int map_add_dev(Ctx *c, const float4 *d_pts, int n, int downsample_on, int *out_added);  // d_pts: device memory
int map_add_pair_dev(Ctx *c, const float4 *d_pts, int m_ds, int m_plain, int *out_added);  // [0,m_ds): down-sampled add, rest: plain add
int map_incremental(Ctx *c, const malio_state_t *state_point, int flg_EKF_inited, const float *h_world_normal_y,
                    int *out_counts);
int map_incremental_select(Ctx *c, const malio_state_t *state_point, int flg_EKF_inited, const float *h_world_normal_y,
                           malio_point_t *out_pts, int *out_index, int cap, int *out_counts2);
// measure.hip: PointToAdd / PointNoNeedDownsample membership + world points, all in ORIGINAL scan order
int mapinc_classify(Ctx *c, const malio_state_t *state_point, int flg_EKF_inited, const float *d_wny, u32 *d_addf,
                    u32 *d_nonf, float4 *d_wp);
int far_knn5(Ctx *c, u32 *d_far);  // measure.hip: unrestricted 5-NN of the queries with nfound < 5, [5][N] sorted order
int map_delete_boxes(Ctx *c, const malio_box_t *boxes, int nb, int *out_deleted);
int map_rebuild_search(Ctx *c);  // neighbour lists of both levels from d_map_in[map_n], now
int map_sync_search(Ctx *c);     // ... only if a mutator left them stale (called by every search entry point)
int map_apply_finish(Ctx *c);    // read the verdict of a queued in-place list update (map_update.hip)

// decode.hip
int decode_livox(Ctx *c, const unsigned char *rec, int n_rec, int n_scans, int pfn, double blind, int eof_point,
                 malio_point_t *out, int cap, int *out_n, double *maximum_time);
int decode_ouster(Ctx *c, const unsigned char *rec, int n, int pfn, double blind, float time_unit_scale, malio_point_t *out,
                  int cap, int *out_n, double *maximum_time);

int decode_velodyne(Ctx *c, const unsigned char *data, int n, const malio_pc2_layout_t &lay, int pfn, double blind,
                    float time_unit_scale, malio_point_t *out, int cap, int *out_n, double *maximum_time);

// voxel.hip
int radix_sort_pairs_u32(Ctx *c, ArenaScope &sc, u32 *&k1, u32 *&k2, u32 *&v1, u32 *&v2, int n, int bits);  // voxel.hip
int voxel_downsample_dev(Ctx *c, ArenaScope &sc, const float *d_pts, int n, float leaf, int normal_mode, float **d_out,
                         int *out_n, bool *passthrough);
int voxel_downsample(Ctx *c, const malio_point_t *pts, int n, float leaf, int normal_mode, malio_point_t *out, int cap,
                     int *out_n);

// measure.hip
int measure_alloc(Ctx *c);
int pass_stage1(Ctx *c, const malio_state_t *s, int converge, double *d_minmax4_out);
// d_minmax4_in: all-reduced extrema (multi-GPU) or null (single GPU: folded inside k_rows_reduce and published
// to d_mm_out)
int pass_stage2(Ctx *c, const double *d_minmax4_in, double *d_mm_out, double *d_sums_out, bool want_rows,
                const GateArgs *gate = nullptr);  // gate: the last kernel announces its completion (see k_final_reduce)
int reset_pass_state(Ctx *c);  // measure.hip: extrema slots and parity as a fresh handle has them
// measure.hip, the speculating pass (k_pass -> k_final_reduce<16>): may this pass run that way? launch it (arguments by
// value, or reading the device loop's control block; results land in h_res like the three-kernel pass'); afterwards:
// *hit = the guessed extrema were the true ones (else the caller redoes the rows)
bool fuse_eligible(Ctx *c, int converge, bool need_guess = true);
int search_skip_begin(Ctx *c);  // a search pass is about to be queued: 1 = it may keep cached neighbours
int pass_fused(Ctx *c, const malio_state_t *s, int converge, const GateArgs *gate, double *row = nullptr);  // row: [sums | extrema words] (default: the pinned result buffer)
void fuse_note(Ctx *c, bool hit);
int enqueue_pass_fused_dev(Ctx *c, const GateArgs *gate);
int fused_collect(Ctx *c, double *sums_out, bool *hit);
int ensure_gate_buffers(Ctx *c);  // ieskf_dev.hip: pinned sequence words + control block, ticket counter
int gate_words(Ctx *c, volatile int **host_msg, int **dev_msg);
int sums_len(const Ctx *c);
int finish_host(Ctx *c, const double *sums, const double *minmax4, malio_measure_out_t *out);
int nearest_search(Ctx *c, const float4 *d_q, int n, int k, u32 *d_idx /*original map index*/, float *d_d2, int *d_cnt);

// Small results the host needs (counts, a few points) come back through one pinned, device-mapped buffer: kernels
// store into it directly (no copy launch at all), and a copy into pinned memory is queued like a kernel (into pageable
// memory it is staged and blocks); either way one stream synchronisation serves everything. 64 KB; words [0, 64) are used by the map code, the rest by whoever needs a bigger read-back.
constexpr size_t MBOX_WORDS = 16384;
inline hipError_t mbox(Ctx *c, u32 **out, u32 **dev = nullptr) {
  if (!c->h_mbox) {
    hipError_t e = hipHostMalloc((void **)&c->h_mbox, sizeof(u32) * MBOX_WORDS, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&c->d_mbox, c->h_mbox, 0);
    if (e != hipSuccess) return e;
  }
  *out = c->h_mbox;
  if (dev) *dev = c->d_mbox;
  return hipSuccess;
}
constexpr int MBOX_SMALL_SEQ = 13;  // sequence word of k_scan_small_dev (mapinc_small_batch polls it instead of synchronising the stream)
constexpr int MBOX_APPLY_SEQ = 14;  // sequence word of k_publish_states (map_apply_finish waits for it, not for the stream)

constexpr int FUSE_COOLDOWN_MIN = 3, FUSE_COOLDOWN_MAX = 48;  // passes without speculation after a wrong guess (fused_collect)

// host/predict.cpp
int predict_step(int L, malio_state_t *x, double *P, double dt, const double *Q, const double *acc, const double *gyro);
// host/ieskf.cpp
int ieskf_update(Ctx *c, malio_xchg_t xchg, malio_state_t *x, double *P, double R, int *stats, double *solve_time);
// csrc/ieskf_dev.hip: the same update as ONE enqueued chain of kernels with the n x n algebra on the device; returns
// MALIO_SMALL_M_FALLBACK untouched inputs when a pass accepted fewer points than there are states (rows path: host loop)
int ieskf_update_device(Ctx *c, malio_state_t *x, double *P, int *stats);
int ieskf_update_device_begin(Ctx *c, const malio_state_t *x, const double *P);  // enqueue everything, return
int ieskf_update_device_end(Ctx *c, malio_state_t *x, double *P, int *stats);    // wait, hand out the results
void free_dev_loop(Ctx *c);
int ieskf_update_gated(Ctx *c, malio_xchg_t xchg, malio_state_t *x, double *P, int *stats, double *solve_time);  // see ieskf_dev.hip
// measure.hip: the pass kernels of one iteration of the device loop (k_search/k_reuse by the control block's converge
// flag, k_rows_reduce, k_final_reduce), all reading their state from c->d_loop
int enqueue_pass_dev(Ctx *c, double *d_sums_out, double *d_mm_out, const GateArgs *gate = nullptr);  // gate: rides on the last kernel
int prepare_scan_dev(Ctx *c, const malio_state_t *s);  // map lists in sync, scan sorted
void publish_pack_now(Ctx *c);  // a one-wave kernel that publishes k_pack_raw's counts (when no grouping kernel will)
int resolve_scan_segments(Ctx *c);  // the per-LiDAR segments of a scan that was packed on the device
void fill_quat_const(const Ctx *c, const malio_state_t *s, QuatConst &qc);
void fill_pass_const(const Ctx *c, const malio_state_t *s, PassConst &pc);
// the same loop over any measurement pass (host/node.cpp drives several GPUs through it); rows_pass == nullptr: no rows
// path, fewer accepted points than states -> MALIO_SMALL_M_FALLBACK
using PassFn = std::function<int(const malio_state_t *, int, malio_measure_out_t *)>;
int ieskf_update_fn(const malio_params_t &prm, const PassFn &pass, const PassFn *rows_pass, int Nscan, void (*hook)(int, void *),
                    void *hook_user, malio_state_t *x, double *P, double R, int *stats, double *solve_time);
int ieskf_step(int L, int maximum_iter, double limit, int i, malio_state_t *x, const malio_state_t *x_propagated,
               const double *P_prop, const double *HtRinvH, const double *HtRinvh, int *t_io, int *converge_out,
               int *done_out, double *P_out);
// the same iteration in two halves: what depends on the iterate alone (dx, the projected P_propagated and its inverse) and
// what needs the pass' normal equations
struct StepPre {
  std::vector<double> P_, Pinv, dx, dx_new;
  int inv_state = 0;  // 0: Pinv not computed, 1: valid, -1: P_ is singular
};
void ieskf_step_pre(int L, const malio_state_t *x, const malio_state_t *x_propagated, const double *P_prop, StepPre &pre,
                    bool with_inverse);
int ieskf_step_post(int L, int maximum_iter, double limit, int i, malio_state_t *x, const malio_state_t *x_propagated,
                    StepPre &pre, const double *HtRinvH, const double *HtRinvh, int *t_io, int *converge_out, int *done_out,
                    double *P_out);

// host/node_exchange.cpp
bool xchg_word_agrees(malio_xchg_t x, int word);

// profiling helpers
void prof_begin(Ctx *c);
void prof_mark(Ctx *c, const char *name);  // records an event AFTER the kernel named `name`
void prof_end(Ctx *c);

}  // namespace malio

struct malio_ctx : malio::Ctx {};
