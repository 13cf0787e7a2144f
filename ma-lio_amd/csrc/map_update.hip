// Map maintenance on the GPU: the two ikd-Tree mutators the mapping loop calls once per scan.
//   Add_Points(PointToAdd, downsample_on)   ikd_Tree.cpp:478-584  (laserMapping.cpp:443-444)
//   Delete_Point_Boxes(cub_needrm)          ikd_Tree.cpp:643-669  (laserMapping.cpp:223)
//
// The reference walks the new points ONE BY ONE; each step reads the tree state the previous steps left.
// That dependency only exists between points of the same downsample voxel, so the work is sharded by voxel:
// one thread replays, in input order, the steps of the new points that fall into its voxel against the stored
// points of that voxel, and every voxel runs in parallel. Outputs are two flag arrays (stored point deleted /
// new point kept). They are applied IN PLACE: a deleted point's slot in the map array and its 27 entries per
// neighbour-list level become tombstones (x = +inf: at infinite distance from everything), kept new points are
// appended to the array and to the slack every list was built with (new cells get a list carved from the tail).
// Map indices therefore stay stable between full rebuilds; a rebuild (compaction + both list levels from scratch)
// happens lazily, at the next search, only when a list, the tail or the directory runs out of room or a fifth of
// the entries are tombstones.
//
// Keeper rule of one step (ikd_Tree.cpp:504-528), restated order-free. With p the new point, S the stored
// points inside the voxel box, "near" meaning calc_dist(q, mid) < downsample_size/8 (a SQUARED distance compared
// with a length: the reference's quirk, kept):
//   - some point of {p} + S is near  ->  the near point of lowest normal_y wins
//   - otherwise                      ->  the point closest to the voxel centre wins
// The reference's loop resolves exact ties by visiting order (p first, then S in tree-traversal order, which is
// not reproducible outside the tree); here ties go to p, then to the lowest map index. Then
// (ikd_Tree.cpp:529-537): if |S| > 1 or the winner coincides with p (same_point, 1e-6 per axis), the box is
// emptied and the winner inserted; otherwise nothing changes.
#include "malio_internal.hpp"

namespace malio {

namespace {

__device__ __forceinline__ bool vox_find(const Cell *__restrict__ table, u32 tmask, u64 key, u32 &start, u32 &count) {
  u32 s = hash_key(key) & tmask;
  while (true) {
    Cell e = table[s];
    if (e.key == key) {
      start = e.start, count = e.count & NL_COUNT;  // (a level-1 neighbour list's count carries NL_SORTED)
      return true;
    }
    if (e.key == EMPTY_KEY) return false;
    s = (s + 1) & tmask;
  }
}

// ikd_Tree.cpp:1694-1699 (float, left to right; the build has -ffp-contract=off)
__device__ __forceinline__ float calc_dist3(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy) + dz * dz;
}

// (the map array is kept in cell order, see map_rebuild_search)
// A slot's rank in insertion order: the side array for the slots the last rebuild sorted, the slot itself for what was
// appended since.
__device__ __forceinline__ u32 map_ord(const u32 *__restrict__ ord, u32 nsorted, u32 slot) { return slot < nsorted ? ord[slot] : slot; }

struct Winner {
  float x, y, z, cov, dist;
  u32 rank;  // 0 = the new point, 1 + rank in insertion order = stored point, 0xFFFFFFFF = the new point kept earlier in this call
  u32 slot;  // a stored point's slot in the map array
  bool near;
};

// does candidate b displace the current winner a?
__device__ __forceinline__ bool displaces(const Winner &a, const Winner &b) {
  if (a.near != b.near) return b.near;
  if (a.near) return b.cov < a.cov || (b.cov == a.cov && b.rank < a.rank);
  return b.dist < a.dist || (b.dist == a.dist && b.rank < a.rank);
}

// One thread per occupied voxel of the NEW points. The stored points of the voxel are found in the level-1
// neighbour list of the new point's cell: the list holds every map point of the 3x3x3 block of cells (edge cf >=
// downsample size) around it, so the whole voxel box is covered; the literal box test decides.
__global__ void __launch_bounds__(BLK) k_vox_add(const Cell *__restrict__ ntable, u32 ntsize,
                                                 const u32 *__restrict__ norig, const float4 *__restrict__ newp,
                                                 const Cell *__restrict__ ltable, u32 ltmask,
                                                 const float4 *__restrict__ lpts, float linv_cf,
                                                 const float4 *__restrict__ mapp, int have_map, float ds,
                                                 unsigned char *del, u32 *dlist, u32 *addf,
                                                 u32 *counters /*[0] adds [1] deletions*/,
                                                 const u32 *__restrict__ ord, u32 nsorted /* map_ord: "lowest map index" = first inserted */) {
  u32 slot = blockIdx.x * BLK + threadIdx.x;
  if (slot >= ntsize) return;
  Cell nc = ntable[slot];
  if (nc.key == EMPTY_KEY || nc.count == 0) return;
  const float near_th = ds / 8;  // ikd_Tree.cpp:510
  const u32 NONE = 0xFFFFFFFFu;
  u32 alive = NONE;  // new point of this voxel currently in the map
  u32 last = 0, adds = 0;
  for (u32 step = 0; step < nc.count; step++) {
    u32 cur = NONE;  // next new point in input order
    for (u32 j = 0; j < nc.count; j++) {
      u32 o = norig[nc.start + j];
      if ((step == 0 || o > last) && o < cur) cur = o;
    }
    last = cur;
    float4 p = newp[cur];
    // ikd_Tree.cpp:494-502: box and centre of the voxel, with the reference's float/double mix
    float bmin[3], bmax[3], mid[3];
    const float pv[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int a = 0; a < 3; a++) {
      bmin[a] = (float)(floor((double)(pv[a] / ds)) * (double)ds);
      bmax[a] = bmin[a] + ds;
      mid[a] = (float)((double)bmin[a] + (double)(bmax[a] - bmin[a]) / 2.0);
    }
    // stored candidates: the level-1 list of the cell of the voxel centre (any point of the box would do)
    u32 es = 0, ec = 0;
    if (have_map) {
      u64 key = cell_key((int)floorf(mid[0] * linv_cf), (int)floorf(mid[1] * linv_cf), (int)floorf(mid[2] * linv_cf));
      vox_find(ltable, ltmask, key, es, ec);
    }
    Winner w;
    w.x = p.x, w.y = p.y, w.z = p.z, w.cov = p.w, w.rank = 0, w.slot = 0xFFFFFFFFu;
    w.dist = calc_dist3(p.x, p.y, p.z, mid[0], mid[1], mid[2]);
    w.near = w.dist < near_th;
    u32 inbox = 0;
    u32 cand[8];  // stored points inside the box (a 0.5 m voxel rarely holds more than two)
    int nc = 0;
    bool cand_overflow = false;
#ifndef VOX_VB
#define VOX_VB 16
#endif
    constexpr int VB = VOX_VB;  // independent loads in flight: the list is ~45 entries from L2/HBM, one voxel per thread and few
                            // threads, so the kernel is as long as its chain of load round trips (4 at a time: 32 us)
    for (u32 j0 = 0; j0 < ec; j0 += VB) {
      float4 q4[VB];
#pragma unroll
      for (int u = 0; u < VB; u++) q4[u] = lpts[(size_t)es + min(j0 + (u32)u, ec - 1)];
#pragma unroll
      for (int u = 0; u < VB; u++) {
        const float4 q = q4[u];  // x = +inf for a tombstone: fails the box test
        // Search_by_range / Delete_by_range leaf test (ikd_Tree.cpp:1263-1274, :807): min <= q < max
        if (j0 + (u32)u >= ec ||
            !(bmin[0] <= q.x && bmax[0] > q.x && bmin[1] <= q.y && bmax[1] > q.y && bmin[2] <= q.z && bmax[2] > q.z))
          continue;
        const u32 mi = __float_as_uint(q.w);
        if (del[mi]) continue;
        inbox++;
        if (nc < 8)
          cand[nc++] = mi;
        else
          cand_overflow = true;
        Winner b;
        b.x = q.x, b.y = q.y, b.z = q.z, b.cov = mapp[mi].w, b.rank = 1u + map_ord(ord, nsorted, mi), b.slot = mi;
        b.dist = calc_dist3(q.x, q.y, q.z, mid[0], mid[1], mid[2]);
        b.near = b.dist < near_th;
        if (displaces(w, b)) w = b;
      }
    }
    if (alive != NONE) {
      float4 q = newp[alive];
      if (bmin[0] <= q.x && bmax[0] > q.x && bmin[1] <= q.y && bmax[1] > q.y && bmin[2] <= q.z && bmax[2] > q.z) {
        inbox++;
        Winner b;
        b.x = q.x, b.y = q.y, b.z = q.z, b.cov = q.w, b.rank = NONE, b.slot = 0xFFFFFFFFu;
        b.dist = calc_dist3(q.x, q.y, q.z, mid[0], mid[1], mid[2]);
        b.near = b.dist < near_th;
        if (displaces(w, b)) w = b;
      }
    }
    // ikd_Tree.cpp:1688-1691
    bool same = fabs((double)(p.x - w.x)) < 1e-6 && fabs((double)(p.y - w.y)) < 1e-6 && fabs((double)(p.z - w.z)) < 1e-6;
    if (inbox > 1 || same) {
      if (!cand_overflow) {
        for (int k = 0; k < nc; k++) {
          const u32 mi = cand[k];
          if (w.slot == mi) continue;
          del[mi] = 1;
          dlist[atomicAdd(&counters[1], 1u)] = mi;  // order is irrelevant: every entry is tombstoned independently
        }
      } else {
        for (u32 j = 0; j < ec; j++) {
          const float4 q = lpts[(size_t)es + j];
          if (!(bmin[0] <= q.x && bmax[0] > q.x && bmin[1] <= q.y && bmax[1] > q.y && bmin[2] <= q.z && bmax[2] > q.z))
            continue;
          const u32 mi = __float_as_uint(q.w);
          if (del[mi] || w.slot == mi) continue;
          del[mi] = 1;
          dlist[atomicAdd(&counters[1], 1u)] = mi;
        }
      }
      if (alive != NONE && w.rank != NONE) {
        float4 q = newp[alive];
        if (bmin[0] <= q.x && bmax[0] > q.x && bmin[1] <= q.y && bmax[1] > q.y && bmin[2] <= q.z && bmax[2] > q.z) {
          addf[alive] = 0;
          alive = NONE;
        }
      }
      if (w.rank == 0) {
        addf[cur] = 1;
        alive = cur;
      }
      adds++;
    }
  }
  if (adds) atomicAdd(&counters[0], adds);
}

__global__ void __launch_bounds__(BLK) k_box_delete(const float4 *__restrict__ mapp, int n,
                                                    const malio_box_t *__restrict__ boxes, int nb, u32 *dlist,
                                                    u32 *counter) {
  int i = blockIdx.x * BLK + threadIdx.x;
  bool hit = false;
  if (i < n) {
    float4 q = mapp[i];  // a deleted slot has x = +inf: inside no box
    for (int b = 0; b < nb; b++) {
      const malio_box_t bx = boxes[b];
      hit = hit || (bx.vertex_min[0] <= q.x && bx.vertex_max[0] > q.x && bx.vertex_min[1] <= q.y &&
                    bx.vertex_max[1] > q.y && bx.vertex_min[2] <= q.z && bx.vertex_max[2] > q.z);
    }
  }
  // one counter bump per wave, every hit takes its slot in the list of deleted indices
  unsigned long long m = __ballot(hit);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  u32 base = 0;
  if (lane == __ffsll((long long)m) - 1) base = atomicAdd(counter, (u32)__popcll(m));
  base = __shfl(base, __ffsll((long long)m) - 1);
  if (hit) dlist[base + __popcll(m & ((1ull << lane) - 1))] = (u32)i;
}

__global__ void __launch_bounds__(BLK) k_map_kill_list(float4 *mapp, const u32 *__restrict__ dlist, int ndel, unsigned char *del,
                                                       u32 del_n) {
  int d = blockIdx.x * BLK + threadIdx.x;
  if (d < ndel) {
    const u32 mi = dlist[d];
    mapp[mi].x = INFINITY;  // the slot stays (indices are stable between rebuilds)
    if (mi < del_n) del[mi] = 0;  // the voxel update's mark: the array is all zero again when the batch is through
  }
}
// keep flags of one Add_Points pair: [0, m_ds) decided by k_vox_add (0), [m_ds, m) kept (1), then a zero and the kernel's
// two counters
__global__ void __launch_bounds__(BLK) k_init_addf(u32 *addf, int m_ds, int m) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i < m + 3) addf[i] = (i >= m_ds && i < m) ? 1u : 0u;
}

__global__ void __launch_bounds__(BLK) k_compact(const float4 *__restrict__ src, const u32 *__restrict__ flag,
                                                 const u32 *__restrict__ pos, int n, const u32 *__restrict__ base,
                                                 float4 *dst) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i < n && flag[i]) dst[(base ? *base : 0u) + pos[i]] = src[i];
}

// two stable compactions of one source in one launch (blockIdx.y): map_incremental's PointToAdd | PointNoNeedDownsample
__global__ void __launch_bounds__(BLK) k_compact2(const float4 *__restrict__ src, const u32 *__restrict__ flagA,
                                                  const u32 *__restrict__ posA, float4 *dstA, const u32 *__restrict__ flagB,
                                                  const u32 *__restrict__ posB, float4 *dstB, int n) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  if (blockIdx.y == 0) {
    if (flagA[i]) dstA[posA[i]] = src[i];
  } else {
    if (flagB[i]) dstB[posB[i]] = src[i];
  }
}

// the same for the positions themselves: which scan point each listed point is (malio_map_incremental_select)
__global__ void __launch_bounds__(BLK) k_compact2_idx(const u32 *__restrict__ flagA, const u32 *__restrict__ posA, u32 *dstA,
                                                      const u32 *__restrict__ flagB, const u32 *__restrict__ posB, u32 *dstB, int n) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  if (blockIdx.y == 0) {
    if (flagA[i]) dstA[posA[i]] = (u32)i;
  } else {
    if (flagB[i]) dstB[posB[i]] = (u32)i;
  }
}

// ---- the usual batch of map_incremental (a few thousand new points) without its first read-back ------------------------
// The lengths of PointToAdd / PointNoNeedDownsample stay on the device (posA[n], posB[n]: exclusive scans over n + 1
// flags) until the ONE read-back that also brings the keeper rule's counts: the compaction, the voxel grouping and the
// scan of the keep flags read them there. The grouping - five launches of group_by_cell for ~1.7 k points - is one
// workgroup with its hash table in LDS. A batch above SMALL_CAP leaves everything untouched (empty voxel table: k_vox_add
// does nothing) and the host takes the general path.
constexpr int SMALL_CAP = 4096;  // new points (both lists)
constexpr int SMALL_TS = 8192;   // voxel table slots (load <= 0.5)
__global__ void __launch_bounds__(BLK) k_compact2_dev(const float4 *__restrict__ src, const u32 *__restrict__ flagA,
                                                      const u32 *__restrict__ posA, const u32 *__restrict__ flagB,
                                                      const u32 *__restrict__ posB, float4 *dst, int n, u32 cap) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  const u32 na = posA[n], nb = posB[n];
  if (na + nb > cap) return;
  if (blockIdx.y == 0) {
    if (flagA[i]) dst[posA[i]] = src[i];
  } else {
    if (flagB[i]) dst[na + posB[i]] = src[i];
  }
}
// one workgroup: voxel table (key -> start, count) + member lists of the first *cntA points of newp (the down-sampled
// list), as group_by_cell leaves them; keep flags as k_init_addf leaves them (the two counters at addf + SMALL_CAP + 1);
// info = {m_ds, m, overflow}
__global__ void __launch_bounds__(1024) k_group_small(const float4 *__restrict__ newp, const u32 *__restrict__ cntA,
                                                      const u32 *__restrict__ cntB, float ds, Cell *table, u32 *orig,
                                                      u32 *addf, u32 *info, u32 cap /* <= SMALL_CAP */) {
  __shared__ u64 s_key[SMALL_TS];
  __shared__ u32 s_cnt[SMALL_TS], s_start[SMALL_TS], s_wsum[16];
  const int t = threadIdx.x;
  const u32 na = *cntA, nb = *cntB;
  const bool over = na + nb > cap;
  const u32 m_ds = over ? 0u : na, m = over ? 0u : na + nb;
  for (int s = t; s < SMALL_TS; s += 1024) s_key[s] = EMPTY_KEY, s_cnt[s] = 0u;
  for (u32 i = t; i < (u32)SMALL_CAP + 3u; i += 1024) addf[i] = (i >= m_ds && i < m) ? 1u : 0u;
  __syncthreads();
  u32 slot[SMALL_CAP / 1024], rank[SMALL_CAP / 1024];
#pragma unroll
  for (int k = 0; k < SMALL_CAP / 1024; k++) {
    const u32 i = (u32)t + 1024u * k;
    slot[k] = 0u, rank[k] = 0u;
    if (i < m_ds) {
      const float4 p = newp[i];  // voxel index exactly as ikd_Tree.cpp:494-499 forms it: floor(x / downsample_size)
      const u64 key = cell_key((int)floorf(p.x / ds), (int)floorf(p.y / ds), (int)floorf(p.z / ds));
      u32 s = hash_key(key) & (SMALL_TS - 1);
      while (true) {
        const u64 old = atomicCAS(reinterpret_cast<unsigned long long *>(&s_key[s]), (unsigned long long)EMPTY_KEY,
                                  (unsigned long long)key);
        if (old == EMPTY_KEY || old == key) break;
        s = (s + 1) & (SMALL_TS - 1);
      }
      slot[k] = s, rank[k] = atomicAdd(&s_cnt[s], 1u);
    }
  }
  __syncthreads();
  {  // exclusive scan of the slot counts: 8 consecutive slots per thread, wave scans, 16 wave totals
    constexpr int PER = SMALL_TS / 1024;
    u32 v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) v[k] = s_cnt[t * PER + k], sum += v[k];
    u32 incl = sum;
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const u32 o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    u32 run = incl - sum;
#pragma unroll
    for (int w = 0; w < 16; w++) run += w < wave ? s_wsum[w] : 0u;
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int s = t * PER + k;
      s_start[s] = run;
      Cell c;
      c.key = s_key[s], c.start = run, c.count = v[k];
      table[s] = c;  // (an empty slot carries EMPTY_KEY and count 0)
      run += v[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SMALL_CAP / 1024; k++) {
    const u32 i = (u32)t + 1024u * k;
    if (i < m_ds) orig[s_start[slot[k]] + rank[k]] = i;
  }
  if (t == 0) info[0] = m_ds, info[1] = m, info[2] = over ? 1u : 0u;
}
// k_scan_small over *m_p + 1 elements (the keep flags and their terminating zero)
__global__ void __launch_bounds__(1024) k_scan_small_dev(const u32 *__restrict__ in, u32 *out, const u32 *__restrict__ m_p,
                                                         u32 *total_out, const u32 *__restrict__ fwd, int nfwd, u32 *seq_word,
                                                         u32 seq) {
  __shared__ u32 wsum[16];
  const int n = (int)*m_p + 1;
  const int per = (n + 1023) / 1024;  // <= 5
  const int base = threadIdx.x * per;
  u32 v[8];
  u32 tsum = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    v[k] = (k < per && base + k < n) ? in[base + k] : 0u;
    tsum += v[k];
  }
  u32 incl = tsum;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  u32 excl = incl - tsum;
#pragma unroll
  for (int w = 0; w < 16; w++) excl += w < wave ? wsum[w] : 0u;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (k < per && base + k < n) {
      out[base + k] = excl;
      if (base + k == n - 1) total_out[0] = excl;
    }
    excl += v[k];
  }
  if ((int)threadIdx.x < nfwd) total_out[1 + threadIdx.x] = fwd[threadIdx.x];
  // the host polls this word (pinned, mapped) instead of synchronising the stream: ~2 us instead of ~10 after the last store
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int ensure_alt(Ctx *c, size_t need) {
  if (need > c->cap_map_alt) {
    if (c->d_map_alt) (void)hipFree(c->d_map_alt);
    c->d_map_alt = nullptr;
    c->cap_map_alt = need + need / 8 + 1024;
    MALIO_HIP(hipMalloc(&c->d_map_alt, sizeof(float4) * c->cap_map_alt));
  }
  return MALIO_OK;
}

void swap_maps(Ctx *c) {
  std::swap(c->d_map_in, c->d_map_alt);
  std::swap(c->cap_map_in, c->cap_map_alt);
}

__global__ void __launch_bounds__(BLK) k_alive_flags(const float4 *__restrict__ mapp, int n, u32 *keep) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i <= n) keep[i] = (i < n && !isinf(mapp[i].x)) ? 1u : 0u;  // keep[n] = 0: the scan leaves the total at [n]
}
// ---- the map array in cell order (round 6) ------------------------------------------------------------------------------
// keep[rank in insertion order] = the slot is alive; keep[n] = 0 (the scan leaves the total there)
__global__ void __launch_bounds__(BLK) k_alive_by_ord(const float4 *__restrict__ mapp, int n, const u32 *__restrict__ ord, u32 nsorted,
                                                      u32 *keep) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i == n) keep[n] = 0u;
  if (i < n) keep[map_ord(ord, nsorted, (u32)i)] = isinf(mapp[i].x) ? 0u : 1u;
}
// the alive slots back in insertion order, packed
__global__ void __launch_bounds__(BLK) k_compact_by_ord(const float4 *__restrict__ mapp, int n, const u32 *__restrict__ ord, u32 nsorted,
                                                        const u32 *__restrict__ kpos, float4 *dst) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  const float4 p = mapp[i];
  if (!isinf(p.x)) dst[kpos[map_ord(ord, nsorted, (u32)i)]] = p;
}
// Sort key: the point's level-1 cell as (cy, cx, cz) - columns of cells, x running fastest: the order the scan's grouping hands
// the queries to the workgroups in (measure.hip: k_sort_count - bucket = column, cy major), so that the 64 queries of a
// workgroup gather their neighbours from ONE stretch of the array. Measured on the CPU (tools/gather_lines.py, BASELINE config
// 2, distinct 128-byte lines of the map array per 64-query workgroup): 80 - 83 in this order, 92 by Morton code, 78 in the
// raster order the synthetic scene happens to be generated in - and 247 for the same map shuffled, which is what a map that
// grew scan by scan looks like to the gather. 10 + 10 + 7 bits (1 152 m x 1 152 m x 144 m; beyond that the key wraps and two
// far-apart cells interleave, which costs those cells' lines their locality and nothing else); ties keep insertion order.
__global__ void __launch_bounds__(BLK) k_cell_keys(const float4 *__restrict__ mapp, int n, float inv_cf, u32 *keys, u32 *vals) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  const float4 p = mapp[i];
  const u32 cx = (u32)(int)floorf(p.x * inv_cf) & 1023u, cy = (u32)(int)floorf(p.y * inv_cf) & 1023u, cz = (u32)(int)floorf(p.z * inv_cf) & 127u;
  keys[i] = (cy << 17) | (cx << 7) | cz;
  vals[i] = (u32)i;
}
__global__ void __launch_bounds__(BLK) k_map_permute(const float4 *__restrict__ src, const u32 *__restrict__ vals, int n, float4 *dst, u32 *ord) {
  int j = blockIdx.x * BLK + threadIdx.x;
  if (j >= n) return;
  const u32 v = vals[j];
  dst[j] = src[v], ord[j] = v;
}
__global__ void __launch_bounds__(BLK) k_fill_u32(u32 *p, u32 v, int n) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void __launch_bounds__(BLK) k_iota_u32(u32 *p, int n) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i < n) p[i] = (u32)i;
}

}  // namespace

// An in-place list update reports through two state words (cells in use, overflow) whether it fitted. map_apply leaves
// that check - the only thing its caller would wait for - to whoever needs the lists next: the mutators return with the
// list maintenance kernels still queued, and the host gets on with the next scan's front end meanwhile.
int map_apply_finish(Ctx *c) {
  if (!c->apply_pending) return MALIO_OK;
  c->apply_pending = false;
  u32 *mb = nullptr;
  MALIO_HIP(mbox(c, &mb));
  // the verdict's sequence word, not the stream: by now the scan upload of the next turn is usually queued behind the list
  // maintenance, and waiting for that too would leave the GPU idle until this thread has launched the scan's grouping
  {
    const volatile u32 *word = mb + MBOX_APPLY_SEQ;
    unsigned long long spins = 0;
    while (__atomic_load_n(const_cast<const u32 *>(word), __ATOMIC_ACQUIRE) != c->apply_seq) {
      if ((++spins & 0x3FFF) == 0) {
        hipError_t q = hipStreamQuery(c->maint_stream && c->maint_enabled ? c->maint_stream : c->stream);
        if (q != hipErrorNotReady && __atomic_load_n(const_cast<const u32 *>(word), __ATOMIC_ACQUIRE) != c->apply_seq) {
          MALIO_HIP(q);
          c->err = "map update: the list maintenance ended without publishing its verdict";
          return MALIO_ERR_HIP;
        }
      }
      __builtin_ia32_pause();
    }
  }
  const u32 *st1 = mb + 16, *st2 = mb + 20;
  c->nl1.ncells = st1[2], c->nl2.ncells = st2[2];
  bool in_place = !(st1[1] || st2[1]);  // a list or the tail region overflowed
  // probing stays short below load 0.7; past it the next search rebuilds with a larger directory
  if ((size_t)st1[2] * 10 > (size_t)(c->nl1.tmask + 1) * 7 || (size_t)st2[2] * 10 > (size_t)(c->nl2.tmask + 1) * 7)
    in_place = false;
  if (!in_place)
    c->search_dirty = true;
  else
    c->n_inplace++;
  return MALIO_OK;
}

int maint_join(Ctx *c) {
  if (!c->maint_pending) return MALIO_OK;
  c->maint_pending = false;
  MALIO_HIP(hipStreamWaitEvent(c->stream, c->ev_maint_done, 0));
  return MALIO_OK;
}
int maint_scope_begin(Ctx *c) {
  if (c->maint_inflight) {
    MALIO_HIP(hipEventSynchronize(c->ev_maint_done));  // (long done: it was recorded a whole scan ago)
    c->maint_inflight = false;
  }
  delete c->maint_scope;
  c->maint_scope = new ArenaScope(c->arena_maint);
  return MALIO_OK;
}
void maint_destroy(Ctx *c) {
  if (c->maint_stream) (void)hipStreamSynchronize(c->maint_stream);
  delete c->maint_scope;
  c->maint_scope = nullptr;
  c->arena_maint.release_all();
  if (c->maint_stream) {
    (void)hipEventDestroy(c->ev_maint_in), (void)hipEventDestroy(c->ev_maint_done);
    (void)hipStreamDestroy(c->maint_stream);
    c->maint_stream = nullptr;
  }
}
// Error paths only: k_vox_add's marks back to zero. Maintenance kernels of the failed batch may still be writing them on the
// maintenance stream (with or without an event recorded behind them): wait for that stream first - an error path may block.
static void clear_marks_now(Ctx *c) {
  if (!c->d_del) return;
  if (c->maint_stream) (void)hipStreamSynchronize(c->maint_stream);
  (void)hipMemsetAsync(c->d_del, 0, c->cap_del, c->stream);
}
// the stream map_apply launches on: its own, entered behind everything queued on `stream` so far.
// inputs_ready = true skips the event dependency on `stream`. INVARIANT the caller then vouches for: everything the
// maintenance kernels read (dlist, d_new, keep, rank, the marks, the totals) was written by kernels of `stream` that the HOST
// has already seen complete - a stream synchronisation, or the sequence word their LAST kernel stores behind a
// __threadfence_system() (mapinc_small_batch: k_scan_small_dev) - and NOTHING has been queued on `stream` since that the
// maintenance kernels depend on. Anything added between that point and map_apply must go back to inputs_ready = false.
static int maint_enter(Ctx *c, hipStream_t *out, bool inputs_ready) {
  if (!c->maint_enabled) {  // MALIO_OPT_MAINT_STREAM
    *out = c->stream;
    return MALIO_OK;
  }
  if (!c->maint_stream) {
    MALIO_HIP(hipStreamCreateWithFlags(&c->maint_stream, hipStreamNonBlocking));
    MALIO_HIP(hipEventCreateWithFlags(&c->ev_maint_in, hipEventDisableTiming));
    MALIO_HIP(hipEventCreateWithFlags(&c->ev_maint_done, hipEventDisableTiming));
  }
  if (!inputs_ready) {  // (an event record + a stream wait cost this thread ~10 us: skipped when the caller has just
                        // synchronised `stream` - the read-back of the counts - and queued nothing since)
    MALIO_HIP(hipEventRecord(c->ev_maint_in, c->stream));
    MALIO_HIP(hipStreamWaitEvent(c->maint_stream, c->ev_maint_in, 0));
  }
  *out = c->maint_stream;
  return MALIO_OK;
}
static int maint_leave(Ctx *c, hipStream_t st) {
  if (st == c->stream) return MALIO_OK;
  MALIO_HIP(hipEventRecord(c->ev_maint_done, st));
  c->maint_pending = true, c->maint_inflight = true;
  return MALIO_OK;
}

int map_sync_search(Ctx *c) {
  if (int rcj = maint_join(c)) return rcj;
  if (int rc = map_apply_finish(c)) return rc;
  if (!c->search_dirty) return MALIO_OK;
  return map_rebuild_search(c);
}

// Full rebuild: sweep the deleted slots out of the map array (indices change), then both list levels from scratch.
int map_rebuild_search(Ctx *c) {
  if (int rcj = maint_join(c)) return rcj;
  if (int rc = map_apply_finish(c)) return rc;
  c->search_dirty = false;
  c->probe_valid = false;  // (the lists move: no cached directory probe survives a rebuild)
  // (1) back to insertion order with the deleted slots swept out (indices change) - nothing to do for an array that IS in
  // insertion order and has no dead slot (a fresh malio_map_build)
  if ((c->map_dead > 0 || c->map_sorted_n > 0) && c->map_n > 0) {
    ArenaScope sc(c->arena);
    u32 *keep = nullptr, *kpos = nullptr, *tiles = nullptr;
    const int n0 = c->map_n;
    MALIO_HIP(sc.get(&keep, (size_t)n0 + 1));
    MALIO_HIP(sc.get(&kpos, (size_t)n0 + 1));
    MALIO_HIP(sc.get(&tiles, (size_t)(n0 + 1 + 1023) / 1024 + 2));
    hipLaunchKernelGGL(k_alive_by_ord, dim3((n0 + 1 + BLK - 1) / BLK), dim3(BLK), 0, c->stream, c->d_map_in, n0, (const u32 *)c->d_map_ord,
                       (u32)c->map_sorted_n, keep);
    exclusive_scan_u32(c, keep, kpos, tiles, n0 + 1);
    u32 alive = 0;
    MALIO_HIP(hipMemcpyAsync(&alive, kpos + n0, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    MALIO_HIP(hipStreamSynchronize(c->stream));
    int rc = ensure_alt(c, (size_t)alive + (size_t)alive / 4 + 4096);
    if (rc != MALIO_OK) return rc;
    hipLaunchKernelGGL(k_compact_by_ord, dim3((n0 + BLK - 1) / BLK), dim3(BLK), 0, c->stream, c->d_map_in, n0, (const u32 *)c->d_map_ord,
                       (u32)c->map_sorted_n, kpos, c->d_map_alt);
    MALIO_HIP(hipStreamSynchronize(c->stream));
    swap_maps(c);
    c->map_n = (int)alive;
    c->map_dead = 0;
    c->map_sorted_n = 0;
    c->map_epoch++;
  }
  // (2) into cell order (k_cell_keys); the lists below are built from the array as it then stands, so their entries, the
  // neighbour ids of the search passes and the gather all speak of the new slots
  if (c->opt_map_cell_order && c->map_n > 1) {
    ArenaScope sc(c->arena);
    const int n = c->map_n;
    u32 *k1 = nullptr, *k2 = nullptr, *v1 = nullptr, *v2 = nullptr;
    MALIO_HIP(sc.get(&k1, (size_t)n));
    MALIO_HIP(sc.get(&k2, (size_t)n));
    MALIO_HIP(sc.get(&v1, (size_t)n));
    MALIO_HIP(sc.get(&v2, (size_t)n));
    hipLaunchKernelGGL(k_cell_keys, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, c->stream, c->d_map_in, n, 1.0f / c->cell, k1, v1);
    if (int rcs = radix_sort_pairs_u32(c, sc, k1, k2, v1, v2, n, 27)) return rcs;
    int rc = ensure_alt(c, (size_t)n + (size_t)n / 4 + 4096);
    if (rc != MALIO_OK) return rc;
    if ((size_t)n > c->cap_map_ord) {
      if (c->d_map_ord) (void)hipFree(c->d_map_ord);
      c->d_map_ord = nullptr;
      c->cap_map_ord = (size_t)n + (size_t)n / 4 + 4096;
      MALIO_HIP(hipMalloc(&c->d_map_ord, sizeof(u32) * c->cap_map_ord));
    }
    hipLaunchKernelGGL(k_map_permute, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, c->stream, c->d_map_in, (const u32 *)v1, n, c->d_map_alt, c->d_map_ord);
    MALIO_HIP(hipStreamSynchronize(c->stream));  // (the sort's temporaries go back to the arena with this scope)
    swap_maps(c);
    c->map_sorted_n = n;
    c->map_epoch++;
  }
  c->nl_tomb = 0;
  if (c->map_n <= 0) return MALIO_OK;
  c->n_rebuilds++;
  // level 1 pruned to the points within one cell edge of each cell (nl_member) unless the map's voxel filter is so
  // coarse that k_vox_add needs the whole block (a voxel's half diagonal must stay inside the kept reach)
  const bool prune1 = !c->opt_nl_full_blocks /* MALIO_OPT_NL_FULL_BLOCKS */ && (float)c->prm.filter_size_map * 0.8660254f <= 0.95f * NL_REACH * c->cell;
  int rc = build_nlist(c, c->d_map_in, c->map_n, c->cell, c->nl1, prune1, c->opt_nl_sorted != 0 /* MALIO_OPT_NL_SORTED */);
  if (rc == MALIO_OK) rc = build_nlist(c, c->d_map_in, c->map_n, std::max(2.0f * c->cell, 2.25f), c->nl2);
  return rc;
}

// room for `extra` more slots at the end of the map array
static int map_reserve(Ctx *c, size_t extra) {
  const size_t need = (size_t)c->map_n + extra;
  if (need <= c->cap_map_in) return MALIO_OK;
  int rc = ensure_alt(c, need + need / 4 + 4096);
  if (rc != MALIO_OK) return rc;
  if (c->map_n > 0)
    MALIO_HIP(hipMemcpyAsync(c->d_map_alt, c->d_map_in, sizeof(float4) * (size_t)c->map_n, hipMemcpyDeviceToDevice, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  swap_maps(c);
  return MALIO_OK;
}

// the two list-state words the host looks at after an in-place update, into the host's mapped buffer
__global__ void k_publish_states(const u32 *__restrict__ s1, const u32 *__restrict__ s2, u32 *out, u32 *seq_word, u32 seq) {
  out[threadIdx.x] = threadIdx.x < 4 ? s1[threadIdx.x] : s2[threadIdx.x - 4];  // (8 threads: one wave)
  __threadfence_system();
  if (threadIdx.x == 0) __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Apply one batch of changes to the map array and, when they fit, to the neighbour lists in place:
//   dlist[ndel]     -> these slots die (x = +inf), their 27 entries per level become tombstones
//   keep[m] != 0    -> d_new[i] is appended as slot map_n + rank[i] and inserted into 27 lists per level
// Falls back to "lists are stale" (rebuilt by the next search) when a list or the directory runs out of room or
// too many tombstones have piled up.
static int map_apply(Ctx *c, const u32 *dlist, u32 ndel, const float4 *d_new, const u32 *keep, const u32 *rank,
                     int m, u32 nadd, bool inputs_ready) {
  if (int rcj = maint_join(c)) return rcj;         // (map_reserve below copies the map array on `stream`)
  if (int rcf = map_apply_finish(c)) return rcf;  // (the previous batch's verdict decides whether this one goes in place)
  const int hw = c->map_n;
  if (nadd) {
    int rc = map_reserve(c, nadd);
    if (rc != MALIO_OK) return rc;
  }
  hipStream_t ms = nullptr;
  if (int rcm = maint_enter(c, &ms, inputs_ready)) return rcm;
  bool in_place = !c->search_dirty && hw > 0;
  // in place only while the directory is comfortably loaded and tombstones stay below a fifth of the live points
  // (a batch that fills the directory or a list is detected by the kernels themselves and reported as overflow)
  if (in_place) {
    if ((size_t)c->nl1.ncells * 10 > (size_t)(c->nl1.tmask + 1) * 7 || (size_t)c->nl2.ncells * 10 > (size_t)(c->nl2.tmask + 1) * 7)
      in_place = false;
    if ((size_t)(c->nl_tomb + ndel) * 5 > (size_t)(hw - c->map_dead)) in_place = false;
  }
  if (ndel) {
    if (in_place) {
      nl_tombstone(c, ms, c->nl1, c->nl2, c->d_map_in, dlist, (int)ndel);
    }
    if (!(in_place && nadd))  // (otherwise a slice of k_nl_ensure's grid does it)
      hipLaunchKernelGGL(k_map_kill_list, dim3((ndel + BLK - 1) / BLK), dim3(BLK), 0, ms, c->d_map_in, dlist,
                         (int)ndel, c->d_del, (u32)(c->d_del ? c->cap_del : 0));
    c->map_dead += (int)ndel, c->nl_tomb += (int)ndel;
  }
  if (nadd) {
    if (!in_place)
      hipLaunchKernelGGL(k_compact, dim3((m + BLK - 1) / BLK), dim3(BLK), 0, ms, d_new, keep, rank, m,
                         (const u32 *)nullptr, c->d_map_in + hw);
    if (in_place) {
      MapSide side;
      side.mapp = c->d_map_in, side.dlist = dlist, side.ndel = (int)ndel, side.del = c->d_del;
      side.del_n = (u32)(c->d_del ? c->cap_del : 0), side.rank = rank, side.dst = c->d_map_in + hw;
      nl_ensure(c, ms, c->nl1, c->nl2, d_new, keep, m, side);
      nl_append(c, ms, c->nl1, c->nl2, d_new, keep, rank, (u32)hw, m);
      u32 *mb = nullptr, *mbd = nullptr;
      MALIO_HIP(mbox(c, &mb, &mbd));
      if (++c->apply_seq == 0) c->apply_seq = 1;
      hipLaunchKernelGGL(k_publish_states, dim3(1), dim3(8), 0, ms, c->nl1.state, c->nl2.state, mbd + 16,
                         mbd + MBOX_APPLY_SEQ, c->apply_seq);
      c->apply_pending = true;  // verdict read by map_apply_finish
    }
    c->map_n = hw + (int)nadd;
  }
  MALIO_HIP(hipGetLastError());
  if (int rcl = maint_leave(c, ms)) return rcl;
  if ((ndel || nadd)) {
    c->map_epoch++;  // neighbour ids handed out before this call may now name a dead slot
    if (!in_place) c->search_dirty = true;
    else if (!c->apply_pending) c->n_inplace++;
  }
  return MALIO_OK;
}

int map_add(Ctx *c, const float4 *h_pts, int m, int downsample_on, int *out_added) {
  MALIO_HIP(hipSetDevice(c->device));
  if (out_added) *out_added = 0;
  if (m <= 0) return MALIO_OK;
  if (int rcm = maint_scope_begin(c)) return rcm;
  float4 *d_new = nullptr;  // read by map_apply's kernels on the maintenance stream after this call has returned
  MALIO_HIP(c->maint_scope->get(&d_new, (size_t)m));
  MALIO_HIP(hipMemcpyAsync(d_new, h_pts, sizeof(float4) * (size_t)m, hipMemcpyHostToDevice, c->stream));
  return map_add_dev(c, d_new, m, downsample_on, out_added);
}

int map_add_dev(Ctx *c, const float4 *d_new, int m, int downsample_on, int *out_added) {
  // set_downsample_param(filter_size_map_min) is what arms DOWNSAMPLE_SWITCH (ikd_Tree.cpp:486); a non-positive
  // size means it was never armed
  const bool ds_on = downsample_on && (float)c->prm.filter_size_map > 0.f;
  return ds_on ? map_add_pair_dev(c, d_new, m, 0, out_added) : map_add_pair_dev(c, d_new, 0, m, out_added);
}

// Add_Points(d_new[0 .. m_ds), true) followed by Add_Points(d_new[m_ds .. m_ds + m_plain), false) as ONE batch: the
// second call neither reads the map nor is read by the first, so applying both at once leaves the same map (slots in
// the same order) and halves the list maintenance and the host round trips of map_incremental (laserMapping.cpp:443-444).
// out_added = return value of the down-sampling call (0 when there is none, ikd_Tree.cpp:563-583).
int map_add_pair_dev(Ctx *c, const float4 *d_new, int m_ds, int m_plain, int *out_added) {
  MALIO_HIP(hipSetDevice(c->device));
  if (out_added) *out_added = 0;
  const int m = m_ds + m_plain;
  if (m <= 0) return MALIO_OK;
  const float ds = (float)c->prm.filter_size_map;
  ArenaScope sc(c->arena);
  if (!c->maint_scope)
    if (int rcm = maint_scope_begin(c)) return rcm;
  ArenaScope &ms = *c->maint_scope;  // what map_apply's kernels read (maintenance stream): alive until the next mutator
  u32 *addf = nullptr, *apos = nullptr, *tiles = nullptr, *counters = nullptr;
  MALIO_HIP(ms.get(&addf, (size_t)m + 1 + 2));  // keep flags, then the two counters of k_vox_add: one clear for both
  counters = addf + m + 1;
  MALIO_HIP(ms.get(&apos, (size_t)m + 1));
  if (m_ds <= 0) {
    hipLaunchKernelGGL(k_fill_u32, dim3((m + BLK - 1) / BLK), dim3(BLK), 0, c->stream, addf, 1u, m);
    hipLaunchKernelGGL(k_iota_u32, dim3((m + BLK - 1) / BLK), dim3(BLK), 0, c->stream, apos, m);
    return map_apply(c, nullptr, 0, d_new, addf, apos, m, (u32)m, false);
  }
  if (ds > 2.0f * c->cell) {
    c->err = "malio_map_add: filter_size_map larger than twice the level-1 cell edge is not supported";
    return MALIO_ERR_BAD_ARG;
  }
  int rc = map_sync_search(c);  // the voxel lookups below read the level-1 lists
  if (rc != MALIO_OK) return rc;
  const int hw = c->map_n;
  CellGrid &gnew = c->gnew;
  rc = group_by_cell(c, d_new, m_ds, 1.f / ds, gnew, nullptr, ds);
  if (rc != MALIO_OK) return rc;
  // the "deleted by this batch" marks live in a persistent array that is all zero between calls (the kill kernel of
  // map_apply clears the marks it consumes): no megabyte-sized clear per scan
  if ((size_t)hw + 1 > c->cap_del) {
    if (c->d_del) (void)hipFree(c->d_del);
    c->d_del = nullptr, c->cap_del = c->cap_map_in + 1024;
    MALIO_HIP(hipMalloc(&c->d_del, c->cap_del));
    MALIO_HIP(hipMemsetAsync(c->d_del, 0, c->cap_del, c->stream));
  }
  unsigned char *del = c->d_del;
  u32 *dlist = nullptr, *mb = nullptr, *mbd = nullptr;
  hipError_t e = ms.get(&dlist, (size_t)hw + 1);
  if (e == hipSuccess) e = sc.get(&tiles, (size_t)(m + 1 + 1023) / 1024 + 2);
  if (e == hipSuccess) e = mbox(c, &mb, &mbd);
  MALIO_HIP(e);
  hipLaunchKernelGGL(k_init_addf, dim3((m + 3 + BLK - 1) / BLK), dim3(BLK), 0, c->stream, addf, m_ds, m);
  const u32 ntsize = gnew.tmask + 1;
  hipLaunchKernelGGL(k_vox_add, dim3((ntsize + BLK - 1) / BLK), dim3(BLK), 0, c->stream, gnew.table, ntsize, gnew.orig,
                     d_new, c->nl1.table, c->nl1.tmask, c->nl1.pts, c->nl1.inv_cf, c->d_map_in, hw > 0 ? 1 : 0, ds, del,
                     dlist, addf, counters, (const u32 *)c->d_map_ord, (u32)c->map_sorted_n);
  // kept points (both parts) | return value of the down-sampling call | deleted map points: stored into the host's
  // mapped buffer by the scan's last kernel
  u32 *h_tot = mb + 8;
  exclusive_scan_u32(c, addf, apos, tiles, m + 1, mbd + 8, counters, 2);
  e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipGetLastError();
  // From here until map_apply's kill kernel has consumed them, k_vox_add's marks are set in the persistent array: every
  // way out that does not reach that kernel clears them, or the next batch would skip those still-live map points
  // (`if (del[mi]) continue` in k_vox_add) and replay the keeper rule on a voxel with holes.
  auto clear_marks = [&] { clear_marks_now(c); };
  if (e != hipSuccess) {
    clear_marks();
    MALIO_HIP(e);
  }
  if (out_added) *out_added = (int)h_tot[1];
  rc = map_apply(c, dlist, h_tot[2], d_new, addf, apos, m, h_tot[0], true);  // (synchronised above, nothing queued since)
  if (rc != MALIO_OK) clear_marks();  // (after the kill kernel ran this clears zeros)
  return rc;
}

int map_delete_boxes(Ctx *c, const malio_box_t *boxes, int nb, int *out_deleted) {
  MALIO_HIP(hipSetDevice(c->device));
  if (out_deleted) *out_deleted = 0;
  const int hw = c->map_n;
  if (nb <= 0 || hw <= 0) return MALIO_OK;
  if (int rcj = maint_join(c)) return rcj;
  if (int rcm = maint_scope_begin(c)) return rcm;
  ArenaScope sc(c->arena);
  malio_box_t *d_boxes = nullptr;
  u32 *dlist = nullptr, *counter = nullptr;
  MALIO_HIP(sc.get(&d_boxes, (size_t)nb));
  MALIO_HIP(c->maint_scope->get(&dlist, (size_t)hw + 1));
  MALIO_HIP(sc.get(&counter, 1));
  MALIO_HIP(hipMemcpyAsync(d_boxes, boxes, sizeof(malio_box_t) * (size_t)nb, hipMemcpyHostToDevice, c->stream));
  MALIO_HIP(hipMemsetAsync(counter, 0, sizeof(u32), c->stream));
  hipLaunchKernelGGL(k_box_delete, dim3((hw + BLK - 1) / BLK), dim3(BLK), 0, c->stream, c->d_map_in, hw, d_boxes, nb, dlist,
                     counter);
  u32 ndel = 0;
  u32 *mb = nullptr;
  MALIO_HIP(mbox(c, &mb));
  MALIO_HIP(hipMemcpyAsync(mb + 24, counter, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  ndel = mb[24];
  MALIO_HIP(hipGetLastError());
  if (out_deleted) *out_deleted = (int)ndel;
  if (ndel == 0) return MALIO_OK;
  return map_apply(c, dlist, ndel, nullptr, nullptr, nullptr, 0, 0, true);
}

}  // namespace malio

// map_incremental(), laserMapping.cpp:398-446, entirely on the device: selection (measure.hip), stable compaction
// of the two lists in scan order, then the two Add_Points calls of :443-444.
namespace malio {
// map_incremental's usual batch (<= SMALL_CAP new points, map down-sampling on): compaction, voxel grouping, keeper rule
// and the scan of the keep flags queued behind the selection's scans with the list lengths still on the device, then
// the one read-back and map_apply. *done = 0: not applicable or the batch was larger - the caller takes the general path
// (the stream is synchronised and nothing was changed).
static int mapinc_small_batch(Ctx *c, ArenaScope &sc, const float4 *wp, const u32 *addf_sel, const u32 *apos_sel,
                              const u32 *nonf_sel, const u32 *npos_sel, int N, int *na_out, int *nn_out, int *added_out,
                              int *done) {
  *done = 0;
  const float ds = (float)c->prm.filter_size_map;
  u32 *mb = nullptr, *mbd = nullptr;
  MALIO_HIP(mbox(c, &mb, &mbd));
  // MALIO_OPT_MAPINC_SMALL = <cap>: 0 = always the general path, a small cap to exercise the fall-back (tests)
  const u32 cap = (u32)std::min(std::max(c->mapinc_small, 0), SMALL_CAP);
  if (!c->mapinc_small || !(ds > 0.f) || ds > 2.0f * c->cell || c->map_n <= 0) {
    MALIO_HIP(hipStreamSynchronize(c->stream));
    return MALIO_OK;
  }
  int rc = map_sync_search(c);  // the voxel lookups read the level-1 lists
  if (rc != MALIO_OK) return rc;
  const int hw = c->map_n;
  if (!c->d_small_table) {
    MALIO_HIP(hipMalloc(&c->d_small_table, sizeof(Cell) * SMALL_TS));
    MALIO_HIP(hipMalloc(&c->d_small_orig, sizeof(u32) * (SMALL_CAP + 8)));
  }
  if ((size_t)hw + 1 > c->cap_del) {
    if (c->d_del) (void)hipFree(c->d_del);
    c->d_del = nullptr, c->cap_del = c->cap_map_in + 1024;
    MALIO_HIP(hipMalloc(&c->d_del, c->cap_del));
    MALIO_HIP(hipMemsetAsync(c->d_del, 0, c->cap_del, c->stream));
  }
  ArenaScope &ms = *c->maint_scope;
  float4 *d_add = nullptr;
  u32 *addf = nullptr, *apos = nullptr, *dlist = nullptr;
  MALIO_HIP(ms.get(&d_add, (size_t)SMALL_CAP));
  MALIO_HIP(ms.get(&addf, (size_t)SMALL_CAP + 3 + 4));  // keep flags, a zero, k_vox_add's two counters, info[3]
  MALIO_HIP(ms.get(&apos, (size_t)SMALL_CAP + 1));
  MALIO_HIP(ms.get(&dlist, (size_t)hw + 1));
  u32 *counters = addf + SMALL_CAP + 1, *info = addf + SMALL_CAP + 3;
  (void)sc;
  hipLaunchKernelGGL(k_compact2_dev, dim3((N + BLK - 1) / BLK, 2), dim3(BLK), 0, c->stream, wp, addf_sel, apos_sel, nonf_sel,
                     npos_sel, d_add, N, cap);
  hipLaunchKernelGGL(k_group_small, dim3(1), dim3(1024), 0, c->stream, d_add, apos_sel + N, npos_sel + N, ds, c->d_small_table,
                     c->d_small_orig, addf, info, cap);
  hipLaunchKernelGGL(k_vox_add, dim3(SMALL_TS / BLK), dim3(BLK), 0, c->stream, c->d_small_table, (u32)SMALL_TS,
                     c->d_small_orig, d_add, c->nl1.table, c->nl1.tmask, c->nl1.pts, c->nl1.inv_cf, c->d_map_in, 1, ds, c->d_del,
                     dlist, addf, counters, (const u32 *)c->d_map_ord, (u32)c->map_sorted_n);
  if (++c->small_seq == 0) c->small_seq = 1;
  hipLaunchKernelGGL(k_scan_small_dev, dim3(1), dim3(1024), 0, c->stream, addf, apos, info + 1, mbd + 8, counters, 2,
                     mbd + MBOX_SMALL_SEQ, c->small_seq);
  hipError_t e = hipGetLastError();
  {  // the totals' sequence word, not the stream (map_apply_finish does the same)
    const volatile u32 *word = mb + MBOX_SMALL_SEQ;
    unsigned long long spins = 0;
    while (e == hipSuccess && __atomic_load_n(const_cast<const u32 *>(word), __ATOMIC_ACQUIRE) != c->small_seq) {
      if ((++spins & 0x3FFF) == 0) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q != hipErrorNotReady && __atomic_load_n(const_cast<const u32 *>(word), __ATOMIC_ACQUIRE) != c->small_seq)
          e = q == hipSuccess ? hipErrorUnknown : q;  // the stream ended without the word
      }
      __builtin_ia32_pause();
    }
  }
  auto clear_marks = [&] { clear_marks_now(c); };
  if (e != hipSuccess) {
    clear_marks();
    MALIO_HIP(e);
  }
  const int na = (int)mb[0], nn = (int)mb[2];
  if (na + nn > (int)cap) return MALIO_OK;  // (nothing was touched: empty voxel table)
  *na_out = na, *nn_out = nn;
  if (added_out) *added_out = (int)mb[8 + 1];
  *done = 1;
  if (na + nn == 0) return MALIO_OK;
  rc = map_apply(c, dlist, mb[8 + 2], d_add, addf, apos, na + nn, mb[8 + 0], true);
  if (rc != MALIO_OK) clear_marks();
  return rc;
}

// selection of map_incremental (laserMapping.cpp:398-442) on the device: d_add = PointToAdd | PointNoNeedDownsample back to
// back in scan order (arena memory of the caller's scope), their counts; d_idx (optional): the scan index of each
static int mapinc_select_dev(Ctx *c, ArenaScope &sc, const malio_state_t *state_point, int flg_EKF_inited,
                             const float *h_world_normal_y, float4 **d_add_out, int *na_out, int *nn_out, u32 **d_idx_out,
                             bool for_apply, int *added_out = nullptr) {
  const int N = c->N;
  if (N <= 0) return MALIO_ERR_NO_SCAN;
  if (int rcj = maint_join(c)) return rcj;  // (the classification reads the map array)
  float *d_wny = nullptr;
  u32 *addf = nullptr, *nonf = nullptr, *apos = nullptr, *npos = nullptr, *tiles = nullptr, *tiles2 = nullptr;
  float4 *wp = nullptr, *d_add = nullptr, *d_non = nullptr;
  MALIO_HIP(sc.get(&addf, (size_t)N + 1));
  MALIO_HIP(sc.get(&nonf, (size_t)N + 1));
  MALIO_HIP(sc.get(&apos, (size_t)N + 1));
  MALIO_HIP(sc.get(&npos, (size_t)N + 1));
  MALIO_HIP(sc.get(&tiles, (size_t)(N + 1 + 1023) / 1024 + 2));
  MALIO_HIP(sc.get(&tiles2, (size_t)(N + 1 + 1023) / 1024 + 2));
  MALIO_HIP(sc.get(&wp, (size_t)N));
  if (h_world_normal_y) {
    // through the pinned staging buffer: a copy out of the caller's pageable array would be staged by the runtime and
    // block this thread for it (~0.1 ms for 100 k floats)
    MALIO_HIP(sc.get(&d_wny, (size_t)N));
    void *stage = nullptr;
    if (int rcs = host_stage(c, sizeof(float) * (size_t)N, &stage)) return rcs;
    memcpy(stage, h_world_normal_y, sizeof(float) * (size_t)N);
    MALIO_HIP(hipMemcpyAsync(d_wny, stage, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, c->stream));
    c->stage_pending = true;
  }
  u32 *mb = nullptr, *mbd = nullptr;
  MALIO_HIP(mbox(c, &mb, &mbd));
  int rc = mapinc_classify(c, state_point, flg_EKF_inited, d_wny, addf, nonf, wp);  // also zeroes addf[N], nonf[N]
  if (rc != MALIO_OK) return rc;
  // the two list lengths go straight from the scans' last kernels into the host's mapped buffer: no copy launches
  exclusive_scan_u32_pair(c, addf, apos, tiles, mbd + 0, nonf, npos, tiles2, mbd + 2, N + 1);
  if (for_apply && !d_idx_out) {
    // the usual batch: everything up to the keeper rule's counts queued behind the scans, ONE read-back
    int done = 0;
    rc = mapinc_small_batch(c, sc, wp, addf, apos, nonf, npos, N, na_out, nn_out, added_out, &done);
    if (rc != MALIO_OK || done) {
      *d_add_out = nullptr;
      return rc;
    }
  } else {
    MALIO_HIP(hipStreamSynchronize(c->stream));
  }
  const int na = (int)mb[0], nn = (int)mb[2];
  // PointToAdd | PointNoNeedDownsample, back to back (for map_apply: read on the maintenance stream after the return)
  MALIO_HIP((for_apply ? *c->maint_scope : sc).get(&d_add, (size_t)na + (size_t)nn));
  d_non = d_add + na;
  hipLaunchKernelGGL(k_compact2, dim3((N + BLK - 1) / BLK, 2), dim3(BLK), 0, c->stream, wp, addf, apos, d_add, nonf, npos,
                     d_non, N);
  if (d_idx_out) {
    u32 *d_idx = nullptr;
    MALIO_HIP(sc.get(&d_idx, (size_t)na + (size_t)nn));
    hipLaunchKernelGGL(k_compact2_idx, dim3((N + BLK - 1) / BLK, 2), dim3(BLK), 0, c->stream, addf, apos, d_idx, nonf, npos,
                       d_idx + na, N);
    *d_idx_out = d_idx;
  }
  *d_add_out = d_add, *na_out = na, *nn_out = nn;
  return MALIO_OK;
}

int map_incremental(Ctx *c, const malio_state_t *state_point, int flg_EKF_inited, const float *h_world_normal_y,
                    int *out_counts) {
  MALIO_HIP(hipSetDevice(c->device));
  ArenaScope sc(c->arena);
  float4 *d_add = nullptr;
  int na = 0, nn = 0;
  if (int rcm = maint_scope_begin(c)) return rcm;
  int added = 0;
  int rc = mapinc_select_dev(c, sc, state_point, flg_EKF_inited, h_world_normal_y, &d_add, &na, &nn, nullptr, true, &added);
  if (rc != MALIO_OK) return rc;
  // ikdtree.Add_Points(PointToAdd, true); ikdtree.Add_Points(PointNoNeedDownsample, false)   (:443-444)
  if (!d_add) {
    // (the usual batch: applied by mapinc_small_batch)
  } else if ((float)c->prm.filter_size_map > 0.f)
    rc = map_add_pair_dev(c, d_add, na, nn, &added);
  else
    rc = map_add_pair_dev(c, d_add, 0, na + nn, nullptr);
  if (out_counts) out_counts[0] = na, out_counts[1] = nn, out_counts[2] = added;
  return rc;
}

// The selection alone, lists to the host (what a node spreads over its shards): out_pts = PointToAdd | PointNoNeedDownsample
// (x, y, z, normal_y), out_index = the scan index of each, counts2 = the two lengths. The map is not touched.
int map_incremental_select(Ctx *c, const malio_state_t *state_point, int flg_EKF_inited, const float *h_world_normal_y,
                           malio_point_t *out_pts, int *out_index, int cap, int *out_counts2) {
  MALIO_HIP(hipSetDevice(c->device));
  ArenaScope sc(c->arena);
  float4 *d_add = nullptr;
  u32 *d_idx = nullptr;
  int na = 0, nn = 0;
  int rc = mapinc_select_dev(c, sc, state_point, flg_EKF_inited, h_world_normal_y, &d_add, &na, &nn, &d_idx, false);
  if (rc != MALIO_OK) return rc;
  out_counts2[0] = na, out_counts2[1] = nn;
  const int m = na + nn;
  if (m > cap) {
    c->err = "malio_map_incremental_select: output capacity too small";
    return MALIO_ERR_BAD_ARG;
  }
  if (m == 0) return MALIO_OK;
  std::vector<float4> hp((size_t)m);
  std::vector<u32> hi((size_t)m);
  MALIO_HIP(hipMemcpyAsync(hp.data(), d_add, sizeof(float4) * (size_t)m, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipMemcpyAsync(hi.data(), d_idx, sizeof(u32) * (size_t)m, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  for (int k = 0; k < m; k++) {
    malio_point_t p;
    memset(&p, 0, sizeof(p));
    p.x = hp[k].x, p.y = hp[k].y, p.z = hp[k].z, p._pad0 = 1.f, p.normal_y = hp[k].w;
    out_pts[k] = p;
    if (out_index) out_index[k] = (int)hi[k];
  }
  return MALIO_OK;
}
}  // namespace malio
