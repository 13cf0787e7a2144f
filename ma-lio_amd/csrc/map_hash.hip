// GPU-resident map that replaces the ikd-Tree as the 5-NN search structure
// (reference: KD_TREE<PointType>, include/ikd-Tree/ikd_Tree.{h,cpp}; Build :369-397): hashed cell directories
// over NEIGHBOUR LISTS (every point replicated into the lists of the 27 cells whose 3x3x3 block contains it), at
// two cell sizes; plus the counting-sort "group by cell" used to order the scan for locality.
// Layout in HBM: points sorted by cell as float4 (x,y,z,bits(original index)) + the original-order
// float4 (x,y,z,normal_y) array the plane fit gathers from + a compact open-addressing table
// of 16-byte {key,start,count} entries. Cell edge c >= sqrt(5) m so the 27 cells around a query
// contain every map point within the reference's acceptance radius (laserMapping.cpp:587).
#include "malio_internal.hpp"

namespace malio {

// Pass 1: insert every point's cell key into a big scratch table, take a rank inside the cell.
__global__ void __launch_bounds__(BLK) k_gbc_insert(const float4 *__restrict__ pts, int n, float inv_c, float div_c,
                                                    u64 *keys, u32 *cnt, u32 mask, u32 *slot_of, u32 *rank_of,
                                                    u32 *ncells) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  int ix, iy, iz;
  if (div_c > 0.f) {  // voxel index exactly as ikd_Tree.cpp:494-499 forms it: floor(x / downsample_size)
    ix = (int)floorf(p.x / div_c), iy = (int)floorf(p.y / div_c), iz = (int)floorf(p.z / div_c);
  } else {
    ix = (int)floorf(p.x * inv_c), iy = (int)floorf(p.y * inv_c), iz = (int)floorf(p.z * inv_c);
  }
  u64 key = cell_key(ix, iy, iz);
  u32 s = hash_key(key) & mask;
  bool fresh = false;
  while (true) {
    u64 old = __builtin_nontemporal_load(&keys[s]);
    if (old == EMPTY_KEY) {
      old = atomicCAS(&keys[s], EMPTY_KEY, key);
      if (old == EMPTY_KEY) {
        fresh = true;
        break;
      }
    }
    if (old == key) break;
    s = (s + 1) & mask;
  }
  unsigned long long m = __ballot(fresh);
  if (fresh && (threadIdx.x & 63) == (unsigned)__ffsll((long long)m) - 1u) atomicAdd(ncells, (u32)__popcll(m));
  slot_of[i] = s;
  rank_of[i] = atomicAdd(&cnt[s], 1u);
}

// Exclusive scan of u32[n]: tile = 1024 elements per workgroup.
__global__ void __launch_bounds__(BLK) k_scan_tiles(const u32 *__restrict__ in, u32 *out, u32 *tile_sums, int n) {
  __shared__ u32 wsum[BLK / 64];
  int base = blockIdx.x * 1024 + threadIdx.x * 4;
  u32 v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = (base + k < n) ? in[base + k] : 0u;
  u32 t = v[0] + v[1] + v[2] + v[3];
  u32 incl = t;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u32 o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  u32 woff = 0;
  for (int w = 0; w < wave; w++) woff += wsum[w];
  u32 excl = woff + incl - t;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (base + k < n) out[base + k] = excl;
    excl += v[k];
  }
  if (threadIdx.x == BLK - 1) tile_sums[blockIdx.x] = woff + incl;
}
// single workgroup: exclusive scan of the tile sums in place
__global__ void __launch_bounds__(BLK) k_scan_sums(u32 *tile_sums, int ntiles) {
  __shared__ u32 wsum[BLK / 64];
  __shared__ u32 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < ntiles; base += BLK) {
    int i = base + threadIdx.x;
    u32 v = i < ntiles ? tile_sums[i] : 0u;
    u32 incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      u32 o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    u32 c0 = carry;
    if (i < ntiles) tile_sums[i] = c0 + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == BLK - 1) carry = c0 + woff + incl;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(BLK) k_scan_add(u32 *out, const u32 *__restrict__ tile_sums, int n) {
  int base = blockIdx.x * 1024 + threadIdx.x * 4;
  u32 off = tile_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (base + k < n) out[base + k] += off;
}

__global__ void __launch_bounds__(BLK) k_gbc_scatter(const float4 *__restrict__ pts, const u32 *__restrict__ in_orig,
                                                     int n, const u32 *__restrict__ slot_of,
                                                     const u32 *__restrict__ rank_of, const u32 *__restrict__ start,
                                                     float4 *out_pts, u32 *out_orig) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  u32 dst = start[slot_of[i]] + rank_of[i];
  u32 og = in_orig ? in_orig[i] : (u32)i;
  float4 p = pts[i];
  out_pts[dst] = make_float4(p.x, p.y, p.z, __uint_as_float(og));
  out_orig[dst] = og;
}

// Pass 3: move the occupied scratch slots into the compact table the queries use.
__global__ void __launch_bounds__(BLK) k_gbc_compact(const u64 *__restrict__ keys, const u32 *__restrict__ cnt,
                                                     const u32 *__restrict__ start, u32 tbig, Cell *table, u32 tmask) {
  u32 s = blockIdx.x * BLK + threadIdx.x;
  if (s >= tbig) return;
  u64 key = keys[s];
  if (key == EMPTY_KEY) return;
  u32 d = hash_key(key) & tmask;
  while (true) {
    u64 old = atomicCAS(&table[d].key, EMPTY_KEY, key);
    if (old == EMPTY_KEY) break;
    d = (d + 1) & tmask;
  }
  table[d].start = start[s];
  table[d].count = cnt[s];
}

__global__ void __launch_bounds__(BLK) k_fill_u64(u64 *p, u64 v, size_t n) {
  size_t i = (size_t)blockIdx.x * BLK + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void __launch_bounds__(BLK) k_clear_table(Cell *t, u32 n) {
  u32 i = blockIdx.x * BLK + threadIdx.x;
  if (i < n) {
    Cell c;
    c.key = EMPTY_KEY, c.start = 0, c.count = 0;
    t[i] = c;
  }
}

// group_by_cell's three clears in one launch: scratch keys <- EMPTY, counters [tbig + 1] <- 0, compact table <- empty cells
__global__ void __launch_bounds__(BLK) k_gbc_prepare(u64 *keys, u32 *cnt, u32 tbig, Cell *table, u32 tsize) {
  const u32 i = blockIdx.x * BLK + threadIdx.x;
  if (i < tbig) keys[i] = EMPTY_KEY;
  if (i <= tbig) cnt[i] = 0u;
  if (i < tsize) {
    Cell c;
    c.key = EMPTY_KEY, c.start = 0, c.count = 0;
    table[i] = c;
  }
}

static u32 next_pow2(u32 v) {
  u32 p = 1;
  while (p < v) p <<= 1;
  return p;
}

void free_grid(CellGrid &g) {
  if (g.table) (void)hipFree(g.table);
  if (g.pts) (void)hipFree(g.pts);
  if (g.orig) (void)hipFree(g.orig);
  g = CellGrid();
}

// Second (last) kernel of a scan of up to 1024 tiles: every workgroup sums the totals of the tiles before it (<= 1024
// values, one block reduction) instead of waiting for a separate single-workgroup scan of the tile sums. The
// workgroup that owns the last element can hand the grand total (callers scan n = m + 1 flags with a zero at the end,
// so out[n - 1] is the count) and a few more words straight to the host's mapped buffer.
__global__ void __launch_bounds__(BLK) k_scan_add_fused(u32 *out, const u32 *__restrict__ tile_sums, int n, u32 *total_out,
                                                        const u32 *__restrict__ fwd, int nfwd) {
  __shared__ u32 wsum[BLK / 64];
  u32 part = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += BLK) part += tile_sums[t];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
  __syncthreads();
  u32 off = 0;
#pragma unroll
  for (int w = 0; w < BLK / 64; w++) off += wsum[w];
  const int base = blockIdx.x * 1024 + threadIdx.x * 4;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (base + k < n) {
      const u32 v = out[base + k] + off;
      out[base + k] = v;
      if (total_out && base + k == n - 1) total_out[0] = v;
    }
  if (total_out && blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < nfwd) total_out[1 + threadIdx.x] = fwd[threadIdx.x];
}

__global__ void k_publish_total(const u32 *__restrict__ last, const u32 *__restrict__ fwd, int nfwd, u32 *total_out) {
  if (threadIdx.x == 0) total_out[0] = *last;
  if ((int)threadIdx.x < nfwd) total_out[1 + threadIdx.x] = fwd[threadIdx.x];
}

// A small scan (n <= 16 K: the per-scan cell tables and keep flags of the map upkeep) in ONE launch of one workgroup:
// 1024 threads x up to 16 consecutive elements, wave scans by shuffles, 16 wave totals through LDS.
constexpr int SCAN_SMALL_MAX = 16384;
__global__ void __launch_bounds__(1024) k_scan_small(const u32 *__restrict__ in, u32 *out, int n, u32 *total_out,
                                                     const u32 *__restrict__ fwd, int nfwd) {
  __shared__ u32 wsum[16];
  const int per = (n + 1023) / 1024;  // <= 16
  const int base = threadIdx.x * per;
  u32 v[16];
  u32 t = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    v[k] = (k < per && base + k < n) ? in[base + k] : 0u;
    t += v[k];
  }
  u32 incl = t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  u32 woff = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) woff += w < wave ? wsum[w] : 0u;
  u32 excl = woff + incl - t;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    if (k < per && base + k < n) {
      out[base + k] = excl;
      if (total_out && base + k == n - 1) total_out[0] = excl;  // (callers scan m + 1 flags with a zero at the end)
    }
    excl += v[k];
  }
  if (total_out && (int)threadIdx.x < nfwd) total_out[1 + threadIdx.x] = fwd[threadIdx.x];
}

// Two scans of the same length in one pair of launches (blockIdx.y picks the array): map_incremental's two flag arrays.
struct ScanPair {
  const u32 *in[2];
  u32 *out[2], *tiles[2], *total[2];
};
__global__ void __launch_bounds__(BLK) k_scan_tiles2(ScanPair p, int n) {
  __shared__ u32 wsum[BLK / 64];
  const u32 *in = p.in[blockIdx.y];
  u32 *out = p.out[blockIdx.y];
  int base = blockIdx.x * 1024 + threadIdx.x * 4;
  u32 v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = (base + k < n) ? in[base + k] : 0u;
  u32 t = v[0] + v[1] + v[2] + v[3];
  u32 incl = t;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u32 o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  u32 woff = 0;
  for (int w = 0; w < wave; w++) woff += wsum[w];
  u32 excl = woff + incl - t;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (base + k < n) out[base + k] = excl;
    excl += v[k];
  }
  if (threadIdx.x == BLK - 1) p.tiles[blockIdx.y][blockIdx.x] = woff + incl;
}
__global__ void __launch_bounds__(BLK) k_scan_add_fused2(ScanPair p, int n) {
  __shared__ u32 wsum[BLK / 64];
  u32 *out = p.out[blockIdx.y];
  const u32 *tile_sums = p.tiles[blockIdx.y];
  u32 *total_out = p.total[blockIdx.y];
  u32 part = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += BLK) part += tile_sums[t];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
  __syncthreads();
  u32 off = 0;
#pragma unroll
  for (int w = 0; w < BLK / 64; w++) off += wsum[w];
  const int base = blockIdx.x * 1024 + threadIdx.x * 4;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (base + k < n) {
      const u32 v = out[base + k] + off;
      out[base + k] = v;
      if (total_out && base + k == n - 1) total_out[0] = v;
    }
}
int exclusive_scan_u32_pair(Ctx *c, const u32 *inA, u32 *outA, u32 *tilesA, u32 *totalA, const u32 *inB, u32 *outB,
                            u32 *tilesB, u32 *totalB, int n) {
  const int ntiles = (n + 1023) / 1024;
  if (ntiles > 1024) {  // (beyond a million flags: two ordinary scans)
    exclusive_scan_u32(c, inA, outA, tilesA, n, totalA);
    return exclusive_scan_u32(c, inB, outB, tilesB, n, totalB);
  }
  ScanPair p;
  p.in[0] = inA, p.in[1] = inB, p.out[0] = outA, p.out[1] = outB, p.tiles[0] = tilesA, p.tiles[1] = tilesB;
  p.total[0] = totalA, p.total[1] = totalB;
  hipLaunchKernelGGL(k_scan_tiles2, dim3(ntiles, 2), dim3(BLK), 0, c->stream, p, n);
  hipLaunchKernelGGL(k_scan_add_fused2, dim3(ntiles, 2), dim3(BLK), 0, c->stream, p, n);
  return MALIO_OK;
}

// total_out (optional, device-visible, e.g. the mapped mailbox): receives out[n - 1] and then fwd[0 .. nfwd)
int exclusive_scan_u32(Ctx *c, const u32 *d_in, u32 *d_out, u32 *d_tiles, int n, u32 *total_out, const u32 *fwd,
                       int nfwd) {
  if (n <= SCAN_SMALL_MAX) {
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, c->stream, d_in, d_out, n, total_out, fwd, nfwd);
    return MALIO_OK;
  }
  int ntiles = (n + 1023) / 1024;
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(BLK), 0, c->stream, d_in, d_out, d_tiles, n);
  if (ntiles <= 1024) {
    hipLaunchKernelGGL(k_scan_add_fused, dim3(ntiles), dim3(BLK), 0, c->stream, d_out, d_tiles, n, total_out, fwd, nfwd);
    return MALIO_OK;
  }
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(BLK), 0, c->stream, d_tiles, ntiles);
  hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(BLK), 0, c->stream, d_out, d_tiles, n);
  if (total_out) hipLaunchKernelGGL(k_publish_total, dim3(1), dim3(64), 0, c->stream, d_out + (n - 1), fwd, nfwd, total_out);
  return MALIO_OK;
}

int group_by_cell(Ctx *c, const float4 *d_in, int n, float inv_cell, CellGrid &g, const u32 *d_in_orig,
                  float div_cell) {
  if (n <= 0) {
    g.n = 0;
    return MALIO_OK;
  }
  u32 tbig = next_pow2((u32)std::max(1024, 2 * n));
  ArenaScope sc(c->arena);
  u64 *keys = nullptr;
  u32 *cnt = nullptr, *start = nullptr, *slot_of = nullptr, *rank_of = nullptr, *tiles = nullptr, *ncells = nullptr;
  int ntiles = (tbig + 1023) / 1024;
  MALIO_HIP(sc.get(&keys, (size_t)tbig));
  MALIO_HIP(sc.get(&cnt, (size_t)tbig + 1));  // + the cell counter: one clear for both
  ncells = cnt + tbig;
  MALIO_HIP(sc.get(&start, (size_t)tbig));
  MALIO_HIP(sc.get(&slot_of, (size_t)n));
  MALIO_HIP(sc.get(&rank_of, (size_t)n));
  MALIO_HIP(sc.get(&tiles, (size_t)ntiles + 1));
  // the compact table is sized for the worst case (every point in its own cell) instead of reading the cell count
  // back: this runs once per scan on ~2 k points, where a host round trip costs more than clearing a few KB
  u32 tsize = next_pow2(std::max(1024u, 4u * (u32)n));
  if ((size_t)tsize > g.cap_table) {
    if (g.table) (void)hipFree(g.table);
    g.table = nullptr;
    g.cap_table = tsize;
    MALIO_HIP(hipMalloc(&g.table, sizeof(Cell) * g.cap_table));
  }
  // scratch keys, counters (+ the cell counter) and the compact table cleared by ONE launch
  hipLaunchKernelGGL(k_gbc_prepare, dim3((std::max(tbig + 1, tsize) + BLK - 1) / BLK), dim3(BLK), 0, c->stream, keys, cnt,
                     tbig, g.table, tsize);
  int nb = (n + BLK - 1) / BLK;
  hipLaunchKernelGGL(k_gbc_insert, dim3(nb), dim3(BLK), 0, c->stream, d_in, n, inv_cell, div_cell, keys, cnt, tbig - 1,
                     slot_of, rank_of, ncells);
  exclusive_scan_u32(c, cnt, start, tiles, (int)tbig);
  if ((size_t)n > g.cap_pts) {
    if (g.pts) (void)hipFree(g.pts);
    if (g.orig) (void)hipFree(g.orig);
    g.pts = nullptr, g.orig = nullptr;
    g.cap_pts = (size_t)n + (size_t)n / 8 + 1024;
    MALIO_HIP(hipMalloc(&g.pts, sizeof(float4) * g.cap_pts));
    MALIO_HIP(hipMalloc(&g.orig, sizeof(u32) * g.cap_pts));
  }
  hipLaunchKernelGGL(k_gbc_scatter, dim3(nb), dim3(BLK), 0, c->stream, d_in, d_in_orig, n, slot_of, rank_of, start,
                     g.pts, g.orig);
  hipLaunchKernelGGL(k_gbc_compact, dim3((tbig + BLK - 1) / BLK), dim3(BLK), 0, c->stream, keys, cnt, start, tbig,
                     g.table, tsize - 1);
  g.tmask = tsize - 1;
  g.ncells = 0;  // not read back
  g.n = n;
  MALIO_HIP(hipGetLastError());
  return MALIO_OK;
}

// ---- neighbour lists -----------------------------------------------------------------------------------------
// pass A: every point bumps the counters of the 27 fine cells whose 3x3x3 block contains it
__global__ void __launch_bounds__(BLK) k_nl_count(const float4 *__restrict__ pts, int n, float inv_cf, u64 *keys, u32 *cnt,
                                                  u32 mask, u32 *ncells, u32 *overflow, int pruned) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  const float gx = p.x * inv_cf, gy = p.y * inv_cf, gz = p.z * inv_cf;
  int ix = (int)floorf(gx), iy = (int)floorf(gy), iz = (int)floorf(gz);
  for (int dz = -1; dz <= 1; dz++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        if (!nl_member(pruned, gx, gy, gz, ix, iy, iz, dx, dy, dz)) continue;
        u64 key = cell_key(ix + dx, iy + dy, iz + dz);
        u32 s = hash_key(key) & mask;
        int probes = 0;
        bool fresh = false;
        while (true) {
          // plain load first: most probes hit a slot that already holds the key (27 points-per-cell hits per
          // slot), and a 64-bit CAS there is a wasted read-modify-write at L2
          u64 old = __builtin_nontemporal_load(&keys[s]);
          if (old == EMPTY_KEY) {
            old = atomicCAS(&keys[s], EMPTY_KEY, key);
            if (old == EMPTY_KEY) {
              fresh = true;
              break;
            }
          }
          if (old == key) break;
          s = (s + 1) & mask;
          if (++probes > 4096) {
            atomicAdd(overflow, 1u);
            return;
          }
        }
        atomicAdd(&cnt[s], 1u);
        // one counter bump per wave instead of one per new cell (same-address atomics serialise)
        unsigned long long m = __ballot(fresh);
        if (fresh && (threadIdx.x & 63) == (unsigned)__ffsll((long long)m) - 1u) atomicAdd(ncells, (u32)__popcll(m));
      }
}
// pass B: place every point into the 27 lists (cursor = running fill count of the list)
__global__ void __launch_bounds__(BLK) k_nl_fill(const float4 *__restrict__ pts, int n, float inv_cf,
                                                 const u64 *__restrict__ keys, const u32 *__restrict__ start, u32 *cursor,
                                                 u32 mask, float4 *out, int pruned) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  float4 rec = make_float4(p.x, p.y, p.z, __uint_as_float((u32)i));
  const float gx = p.x * inv_cf, gy = p.y * inv_cf, gz = p.z * inv_cf;
  int ix = (int)floorf(gx), iy = (int)floorf(gy), iz = (int)floorf(gz);
  for (int dz = -1; dz <= 1; dz++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        if (!nl_member(pruned, gx, gy, gz, ix, iy, iz, dx, dy, dz)) continue;
        u64 key = cell_key(ix + dx, iy + dy, iz + dz);
        u32 s = hash_key(key) & mask;
        while (keys[s] != key) s = (s + 1) & mask;
        out[(size_t)start[s] + atomicAdd(&cursor[s], 1u)] = rec;
      }
}

void free_nl_scratch(NlScratch &s) {
  if (s.keys) (void)hipFree(s.keys);
  if (s.cnt) (void)hipFree(s.cnt);
  if (s.start) (void)hipFree(s.start);
  if (s.capv) (void)hipFree(s.capv);
  if (s.tiles) (void)hipFree(s.tiles);
  if (s.counters) (void)hipFree(s.counters);
  s = NlScratch();
}

void free_nlist(NList &nl) {
  if (nl.table) (void)hipFree(nl.table);
  if (nl.pts) (void)hipFree(nl.pts);
  if (nl.cap) (void)hipFree(nl.cap);
  if (nl.inc) (void)hipFree(nl.inc);
  if (nl.work) (void)hipFree(nl.work);
  if (nl.state) (void)hipFree(nl.state);
  nl = NList();
}

// capacity of a list built with `cnt` entries: room for a quarter more, at least NL_MIN_SLACK - rounded up to whole 128-byte
// lines (8 entries): capacities are what the lists' starts are the prefix sums of (and what the tail's bump cursor advances
// by), so every list STARTS on a line. A 44-entry list then lies in 6 lines instead of 6.5 on average, and the 64 bytes a
// query's four lanes load per instruction never straddle two - what the search pass is short of is outstanding L1 misses
// per CU (profiles/round5/r05n_tcp_counters.txt), and every line is one.
constexpr u32 NL_MIN_SLACK = 8;
__host__ __device__ inline u32 nl_capacity(u32 cnt) { return (cnt + max(NL_MIN_SLACK, cnt / 4) + 7u) & ~7u; }
constexpr int NL_MAX_PROBES = 1024;  // linear-probe bound of the incremental kernels (a full directory must not hang them)
__global__ void __launch_bounds__(BLK) k_nl_caps(const u32 *__restrict__ cnt, u32 *capv, u32 n) {
  u32 i = blockIdx.x * BLK + threadIdx.x;
  if (i > n) return;  // capv[n] = 0: the exclusive scan then leaves the total there
  u32 c = i < n ? cnt[i] : 0u;
  capv[i] = c ? nl_capacity(c) : 0u;
}
// scratch slots -> compact directory, with the capacity of every list next to it
__global__ void __launch_bounds__(BLK) k_nl_compact(const u64 *__restrict__ keys, const u32 *__restrict__ cnt,
                                                    const u32 *__restrict__ start, const u32 *__restrict__ capv, u32 tbig,
                                                    Cell *table, u32 *cap, u32 tmask) {
  u32 s = blockIdx.x * BLK + threadIdx.x;
  if (s >= tbig) return;
  u64 key = keys[s];
  if (key == EMPTY_KEY) return;
  u32 d = hash_key(key) & tmask;
  while (true) {
    u64 old = atomicCAS(&table[d].key, EMPTY_KEY, key);
    if (old == EMPTY_KEY) break;
    d = (d + 1) & tmask;
  }
  table[d].start = start[s];
  table[d].count = cnt[s];
  cap[d] = capv[s];
}


// ---- level-1 lists sorted by distance from the cell centre (round 5) -------------------------------------------------
// What a search pass is short of is outstanding L1 misses per CU (profiles/round5/r05n_tcp_counters.txt): every 128-byte line
// of a list a query reads is one. A list in ARBITRARY order must be read whole. In order of distance from the centre of its
// cell, a query that has read the first 32 entries knows that everything behind them lies at least as far from the centre
// as the farthest one it has seen, r - and so at least r - |query - centre| from the query (triangle inequality); if the
// fifth distance so far is smaller, the rest cannot change the result (measure.hip: nl_walk<.., EARLY>; 96 % of the queries
// of BASELINE config 2 stop there: 4.1 lines of list per query instead of 6.2).
// One wave sorts one list in registers: up to four entries per lane, every entry's final position = the number of entries
// that sort before it under (distance, position) - n shuffles, no scratch, in place (every lane holds its entries before
// any lane stores). Tombstones (x = +inf: their distance is +inf) go to the end. Lists above NL_SORT_MAX entries are left
// as they are, unflagged. work == nullptr: every list of the directory (after a build); else the lists work[0 .. *nwork)
// (the ones a batch appended to: k_nl_place).
__device__ __forceinline__ float nl_centre_d2(const float4 &e, float cx, float cy, float cz) {
  const float dx = e.x - cx, dy = e.y - cy, dz = e.z - cz;
  const float d2 = dx * dx + dy * dy + dz * dz;  // == measure.hip: nl_walk's r2 of the entries it has read
  // (a non-finite key - a tombstone's +inf, or a NaN coordinate somebody put into the map - sorts behind everything: ranking by
  // comparison needs a total order, and a NaN would compare false both ways and collide with another entry's rank)
  return d2 < INFINITY ? d2 : INFINITY;
}
// one list of n <= 64 K entries, by the whole wave: K entries per lane (the usual list: K = 1)
template <int K>
__device__ __forceinline__ void nl_sort_body(float4 *lst, u32 n, float cx, float cy, float cz, int lane) {
  float4 e[K];
  float key[K];
  u32 rank[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const u32 idx = (u32)lane + 64u * k;
    key[k] = INFINITY, rank[k] = 0;
    if (idx < n) {
      e[k] = lst[idx];
      key[k] = nl_centre_d2(e[k], cx, cy, cz);
    }
  }
#pragma unroll
  for (int kk = 0; kk < K; kk++) {
    if (64u * kk >= n) break;  // (wave-uniform)
    const u32 lim = min(64u, n - 64u * kk);
    for (u32 jj = 0; jj < lim; jj++) {
      // entry j's key to everybody: a scalar (v_readlane with a uniform lane index; a shuffle is an LDS-crossbar round trip
      // per entry and made this kernel 68 us for a batch's ~10 k lists)
      const float kj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(key[kk]), (int)jj));
      const u32 j = 64u * kk + jj;
#pragma unroll
      for (int k = 0; k < K; k++) {
        const u32 idx = (u32)lane + 64u * k;
        rank[k] += (kj < key[k] || (kj == key[k] && j < idx)) ? 1u : 0u;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    const u32 idx = (u32)lane + 64u * k;
    if (idx < n) lst[rank[k]] = e[k];
  }
}
// A list that WAS in order (its first n0 entries) and got entries appended: the tail's entries are ranked among all, the live
// prefix entries keep their order and move up by the number of tail entries that sort before them - a compare and a ballot per
// tail entry instead of a compare per pair (a scan's batch touches ~20 k lists with a few new entries each; the kernel sits
// between a scan's map update and the next scan's first search). Tombstones in the prefix (round 6: the voxel filter of every
// batch replaces map points - dead entries sit in exactly the lists the batch appends to, and sorting those from scratch was
// 15 of the kernel's 26 us): their key is +inf wherever they stand, the LIVE prefix entries are still in order among
// themselves; a dead entry's place is behind every live one, among the dead by position - two ballots, no loop.
template <int K>
__device__ __forceinline__ void nl_merge_body(float4 *lst, u32 n0, u32 n, float cx, float cy, float cz, int lane) {
  float4 e[K];
  float key[K];
  u32 rank[K];
  unsigned long long livem[K], deadm[K];  // (wave-uniform) the entries of slot k with a finite / a non-finite key
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const u32 idx = (u32)lane + 64u * k;
    key[k] = INFINITY, rank[k] = idx;
    if (idx < n) {
      e[k] = lst[idx];
      key[k] = nl_centre_d2(e[k], cx, cy, cz);
    }
    livem[k] = __ballot(idx < n && key[k] < INFINITY);
    deadm[k] = __ballot(idx < n && !(key[k] < INFINITY));
  }
  u32 nlive = 0;
#pragma unroll
  for (int k = 0; k < K; k++) nlive += (u32)__popcll(livem[k]);
  // prefix entries: a live one starts at its place among the live prefix entries, a dead one (prefix or tail: the order under
  // (key, position) puts every +inf key behind the finite ones, by position) gets its final place right away
  {
    u32 live_before = 0, dead_before = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const u32 idx = (u32)lane + 64u * k;
      // (live entries of the PREFIX only: the tail's live entries are ranked by the loop below)
      const unsigned long long pm = 64u * k + 64u <= n0 ? ~0ull : (64u * k >= n0 ? 0ull : (1ull << (n0 - 64u * k)) - 1ull);
      if (idx < n) {
        if (key[k] < INFINITY) {
          if (idx < n0) rank[k] = live_before + (u32)__popcll(livem[k] & pm & below);
        } else {
          rank[k] = nlive + dead_before + (u32)__popcll(deadm[k] & below);
        }
      }
      live_before += (u32)__popcll(livem[k] & pm);
      dead_before += (u32)__popcll(deadm[k]);
    }
  }
  for (u32 t = n0; t < n; t++) {  // (wave-uniform)
    const int tl = (int)(t & 63u);
    float kt = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++)
      if ((t >> 6) == (u32)k) kt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(key[k]), tl));
    if (!(kt < INFINITY)) continue;  // (a dead tail entry has its place already and sorts before nobody alive)
    u32 before = 0;  // entries that sort before entry t: its final position
#pragma unroll
    for (int k = 0; k < K; k++) {
      const u32 idx = (u32)lane + 64u * k;
      const bool live = idx < n;
      before += (u32)__popcll(__ballot(live && (key[k] < kt || (key[k] == kt && idx < t))));
      if (live && idx < n0 && kt < key[k] && key[k] < INFINITY) rank[k] += 1;  // (a tie leaves the older entry in front)
    }
#pragma unroll
    for (int k = 0; k < K; k++)
      if ((u32)lane + 64u * k == t) rank[k] = before;
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    const u32 idx = (u32)lane + 64u * k;
    if (idx < n && rank[k] != idx) lst[rank[k]] = e[k];
  }
}
// (the merge costs ~8 K instructions per tail entry, the sort from scratch ~4 K per entry of the list: a tail up to the prefix's
// length is merged)
// (slot: the list's directory slot, old: its count word before the batch's appends - 0: unknown -, key / start / n: the cell and
// the list as they are now; every lane of the wave calls with the same values)
__device__ __forceinline__ void nl_sort_list(const NlDev &nl, u32 slot, u32 old, u64 ckey, u32 cstart, u32 n, int lane) {
  if (ckey == EMPTY_KEY) return;
  if (n <= 1) {
    if (lane == 0) nl.table[slot].count = n | NL_SORTED;  // (nothing to order)
    return;
  }
  if (n > NL_SORT_MAX) return;  // stays as it is, unflagged (k_nl_append cleared the flag if it had one)
  const u64 B = 1ull << 20;
  const float cx = ((float)((int)(ckey & 0x1FFFFF) - (int)B) + 0.5f) * nl.cf,
              cy = ((float)((int)((ckey >> 21) & 0x1FFFFF) - (int)B) + 0.5f) * nl.cf,
              cz = ((float)((int)((ckey >> 42) & 0x1FFFFF) - (int)B) + 0.5f) * nl.cf;
  float4 *lst = nl.pts + (size_t)cstart;
  static_assert(NL_SORT_MAX == 256, "the dispatch below covers 64 / 128 / 256 entries");
  const u32 n0 = old & NL_COUNT;
  if ((old & NL_SORTED) && n0 >= 1 && n0 < n && n - n0 <= n0) {
    if (n <= 64) nl_merge_body<1>(lst, n0, n, cx, cy, cz, lane);
    else if (n <= 128) nl_merge_body<2>(lst, n0, n, cx, cy, cz, lane);
    else nl_merge_body<4>(lst, n0, n, cx, cy, cz, lane);
  } else {
    if (n <= 64) nl_sort_body<1>(lst, n, cx, cy, cz, lane);
    else if (n <= 128) nl_sort_body<2>(lst, n, cx, cy, cz, lane);
    else nl_sort_body<4>(lst, n, cx, cy, cz, lane);
  }
  if (lane == 0) nl.table[slot].count = n | NL_SORTED;
}
__global__ void __launch_bounds__(BLK) k_nl_sort(NlDev nl, const u32 *__restrict__ work, const u32 *__restrict__ nwork) {
  const int lane = threadIdx.x & 63;
  const u32 wave = (blockIdx.x * BLK + threadIdx.x) >> 6, nwaves = (gridDim.x * BLK) >> 6;
  // a batch's work list: one list per wave and round - unless the batch touched more lists than the list holds (k_nl_place
  // counted them all, stored what fitted): then every list of the directory is put in order, as after a build (0.5 ms at a
  // million points; a batch of more than ~150 k new points - round-5 advisor: such lists used to stay unflagged, walked whole,
  // until the next rebuild)
  if (work && *nwork <= nl.work_cap) {
    const u32 nitems = *nwork;
    for (u32 item = wave; item < nitems; item += nwaves) {
      const uint4 w0 = ((const uint4 *)work)[2 * (size_t)item], w1 = ((const uint4 *)work)[2 * (size_t)item + 1];
      nl_sort_list(nl, w0.x, w0.y, (u64)w1.x | ((u64)w1.y << 32), w0.w, w0.z, lane);
    }
    return;
  }
  // the whole directory: 64 slots per wave and round, most of them empty - the occupied ones one after the other
  const u32 nitems = nl.tmask + 1;
  for (u32 base = wave * 64; base < nitems; base += nwaves * 64) {
    bool todo = false;
    if (base + lane < nitems) {
      const Cell c = nl.table[base + lane];
      todo = c.key != EMPTY_KEY;
    }
    unsigned long long m = __ballot(todo);
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const Cell c = nl.table[base + (u32)src];
      nl_sort_list(nl, base + (u32)src, 0u, c.key, c.start, c.count & NL_COUNT, lane);
    }
  }
}

NlDev nl_dev(const NList &nl);

int build_nlist(Ctx *c, const float4 *d_in, int n, float cf, NList &nl, bool pruned, bool sorted) {
  nl.cf = cf;
  nl.pruned = pruned;
  nl.sorted = sorted;
  nl.inv_cf = 1.0f / nl.cf;
  // scratch table: halo cells are a few times the occupied ones; 8 slots per point keeps the load low
  u32 tbig = next_pow2((u32)std::max(4096, 8 * n));
  int ntiles = (tbig + 1 + 1023) / 1024;
  NlScratch &sc = c->nl_scratch;  // kept across rebuilds
  if (tbig > sc.cap) {
    free_nl_scratch(sc);
    MALIO_HIP(hipMalloc(&sc.keys, sizeof(u64) * tbig));
    MALIO_HIP(hipMalloc(&sc.cnt, sizeof(u32) * (tbig + 1)));
    MALIO_HIP(hipMalloc(&sc.start, sizeof(u32) * (tbig + 1)));
    MALIO_HIP(hipMalloc(&sc.capv, sizeof(u32) * (tbig + 1)));
    MALIO_HIP(hipMalloc(&sc.tiles, sizeof(u32) * (ntiles + 2)));
    MALIO_HIP(hipMalloc(&sc.counters, sizeof(u32) * 2));
    sc.cap = tbig;
  }
  u64 *keys = sc.keys;
  u32 *cnt = sc.cnt, *start = sc.start, *capv = sc.capv, *tiles = sc.tiles, *counters = sc.counters;
  hipLaunchKernelGGL(k_fill_u64, dim3((tbig + BLK - 1) / BLK), dim3(BLK), 0, c->stream, keys, EMPTY_KEY, (size_t)tbig);
  MALIO_HIP(hipMemsetAsync(cnt, 0, sizeof(u32) * tbig, c->stream));
  MALIO_HIP(hipMemsetAsync(counters, 0, sizeof(u32) * 2, c->stream));
  int nb = (n + BLK - 1) / BLK;
  hipLaunchKernelGGL(k_nl_count, dim3(nb), dim3(BLK), 0, c->stream, d_in, n, nl.inv_cf, keys, cnt, tbig - 1, counters,
                     counters + 1, pruned ? 1 : 0);
  // every list gets slack for incremental inserts (map_update.hip); starts = exclusive scan of the capacities
  hipLaunchKernelGGL(k_nl_caps, dim3((tbig + 1 + BLK - 1) / BLK), dim3(BLK), 0, c->stream, cnt, capv, tbig);
  exclusive_scan_u32(c, capv, start, tiles, (int)tbig + 1);
  u32 h_cnt[3] = {0, 0, 0};
  MALIO_HIP(hipMemcpyAsync(h_cnt, counters, sizeof(u32) * 2, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipMemcpyAsync(&h_cnt[2], start + tbig, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  if (h_cnt[1] != 0) {
    c->err = "neighbour-list directory overflow";
    return MALIO_ERR_ALLOC;
  }
  const size_t used = h_cnt[2];
  // tail of the array: lists of cells that do not exist yet are carved from here by incremental inserts
  const size_t want = used + used / 4 + ((size_t)1 << 20);
  if (want > nl.cap_pts || !nl.pts) {
    if (nl.pts) (void)hipFree(nl.pts);
    nl.pts = nullptr;
    nl.cap_pts = want + want / 16;
    if (nl.cap_pts > 0xFFFFFF00ull) {
      c->err = "neighbour lists exceed 2^32 entries";
      return MALIO_ERR_ALLOC;
    }
    // + NL_GUARD entries nobody owns: the pipelined walk (measure.hip, nl_search) reads whole rounds, and the last round of
    // the array's last list may reach past its end
    MALIO_HIP(hipMalloc(&nl.pts, sizeof(float4) * (nl.cap_pts + NL_GUARD)));
  }
  MALIO_HIP(hipMemsetAsync(cnt, 0, sizeof(u32) * tbig, c->stream));  // reuse as the fill cursor
  hipLaunchKernelGGL(k_nl_fill, dim3(nb), dim3(BLK), 0, c->stream, d_in, n, nl.inv_cf, keys, start, cnt, tbig - 1,
                     nl.pts, pruned ? 1 : 0);
  // compact directory (the fill cursors now equal the list lengths); sized for growth to load 0.7
#ifndef NL_DIR_X
#define NL_DIR_X 3u
#endif
  u32 tsize = next_pow2(std::max(1024u, NL_DIR_X * h_cnt[0]));
  if ((size_t)tsize > nl.cap_table || !nl.table) {
    if (nl.table) (void)hipFree(nl.table);
    if (nl.cap) (void)hipFree(nl.cap);
    nl.table = nullptr, nl.cap = nullptr;
    nl.cap_table = tsize;
    MALIO_HIP(hipMalloc(&nl.table, sizeof(Cell) * nl.cap_table));
    MALIO_HIP(hipMalloc(&nl.cap, sizeof(u32) * nl.cap_table));
    if (nl.inc) (void)hipFree(nl.inc);
    MALIO_HIP(hipMalloc(&nl.inc, sizeof(u32) * nl.cap_table));
  }
  if (!nl.state) MALIO_HIP(hipMalloc(&nl.state, sizeof(u32) * 4));
  hipLaunchKernelGGL(k_clear_table, dim3((tsize + BLK - 1) / BLK), dim3(BLK), 0, c->stream, nl.table, tsize);
  MALIO_HIP(hipMemsetAsync(nl.cap, 0, sizeof(u32) * tsize, c->stream));
  MALIO_HIP(hipMemsetAsync(nl.inc, 0, sizeof(u32) * tsize, c->stream));
  hipLaunchKernelGGL(k_nl_compact, dim3((tbig + BLK - 1) / BLK), dim3(BLK), 0, c->stream, keys, cnt, start, capv, tbig,
                     nl.table, nl.cap, tsize - 1);
  const u32 h_state[4] = {(u32)used, 0u, h_cnt[0], 0u};  // bump cursor, overflow flag, cells, -
  MALIO_HIP(hipMemcpyAsync(nl.state, h_state, sizeof(h_state), hipMemcpyHostToDevice, c->stream));
  nl.tmask = tsize - 1, nl.ncells = h_cnt[0], nl.total = used, nl.entries = (size_t)27 * (size_t)n;
  if (sorted)
    hipLaunchKernelGGL(k_nl_sort, dim3(std::min<u32>(8192u, (tsize + BLK - 1) / BLK)), dim3(BLK), 0, c->stream, nl_dev(nl),
                       (const u32 *)nullptr, (const u32 *)nullptr);
  MALIO_HIP(hipStreamSynchronize(c->stream));
  MALIO_HIP(hipGetLastError());
  return MALIO_OK;
}

// ---- incremental maintenance (called by map_update.hip; same stream, never concurrent with a search) -----------
// (1) make sure the 27 cells around every kept new point have a directory entry and a list to append to
// All four kernels serve both list levels in one launch: blockIdx.y picks the level (two launches of latency-bound
// kernels back to back cost twice the latency, one launch of both overlaps them).
__global__ void __launch_bounds__(BLK) k_nl_ensure(const float4 *__restrict__ newp, const u32 *__restrict__ keep, int m,
                                                   NlDev nl_a, NlDev nl_b, MapSide ms) {
  if (blockIdx.y == 2) {
    // the map array's share of the batch, riding along (two launches less; the tombstone kernel, which reads the
    // deleted points' coordinates, is through): deleted slots die, kept new points are appended
    const int stride = (int)gridDim.x * BLK;
    for (int d = blockIdx.x * BLK + threadIdx.x; d < ms.ndel; d += stride) {
      const u32 mi = ms.dlist[d];
      ms.mapp[mi].x = INFINITY;  // the slot stays (indices are stable between rebuilds)
      if (mi < ms.del_n) ms.del[mi] = 0;  // the voxel update's mark: the array is all zero again when the batch is through
    }
    for (int i = blockIdx.x * BLK + threadIdx.x; i < m; i += stride)
      if (keep[i]) ms.dst[ms.rank[i]] = newp[i];
    return;
  }
  const NlDev nl = blockIdx.y ? nl_b : nl_a;
  // 32 lanes per point, one of its 27 cells each
  const long long t = (long long)blockIdx.x * BLK + threadIdx.x;
  if (t == 0) nl.state[3] = 0;  // the batch's work list of touched lists (k_nl_place fills it, k_nl_sort consumes it) starts empty
  const int i = (int)(t >> 5), cidx = (int)(t & 31);
  if (i >= m || cidx >= 27 || !keep[i]) return;
  float4 p = newp[i];
  const float gx = p.x * nl.inv_cf, gy = p.y * nl.inv_cf, gz = p.z * nl.inv_cf;
  int ix = (int)floorf(gx), iy = (int)floorf(gy), iz = (int)floorf(gz);
  if (!nl_member(nl.pruned, gx, gy, gz, ix, iy, iz, cidx % 3 - 1, (cidx / 3) % 3 - 1, cidx / 9 - 1)) return;
  u64 key = cell_key(ix + cidx % 3 - 1, iy + (cidx / 3) % 3 - 1, iz + cidx / 9 - 1);
  u32 s = hash_key(key) & nl.tmask;
  int probes = 0;
  while (true) {
    if (++probes > NL_MAX_PROBES) {  // the directory filled up under this batch: the host rebuilds
      atomicExch(&nl.state[1], 1u);
      return;
    }
    u64 old = __builtin_nontemporal_load(&nl.table[s].key);
    if (old == EMPTY_KEY) {
      old = atomicCAS(&nl.table[s].key, EMPTY_KEY, key);
      if (old == EMPTY_KEY) {  // this thread created the cell; its list is sized and placed by the next kernel
        nl.table[s].start = 0;
        nl.table[s].count = 0;
        nl.cap[s] = 0;
        atomicAdd(&nl.state[2], 1u);
        atomicAdd(&nl.inc[s], 1u);  // (1b) what this batch brings to the cell (was a launch of its own: k_nl_place, mode 0)
        return;
      }
    }
    if (old == key) {
      atomicAdd(&nl.inc[s], 1u);
      return;
    }
    s = (s + 1) & nl.tmask;
  }
}
// (1b) mode 0: how many entries does this batch bring to each touched cell?  (1c) mode 1: one lane per touched cell
//      makes room - nothing to do when the slack suffices, otherwise (new cell, or a list at the map frontier that
//      outgrew its slack) the list moves to the tail region with fresh slack.
__global__ void __launch_bounds__(BLK) k_nl_place(const float4 *__restrict__ newp, const u32 *__restrict__ keep, int m,
                                                  NlDev nl_a, NlDev nl_b, int mode) {
  const NlDev nl = blockIdx.y ? nl_b : nl_a;
  const long long t = (long long)blockIdx.x * BLK + threadIdx.x;
  const int i = (int)(t >> 5), cidx = (int)(t & 31);
  bool live = i < m && cidx < 27 && keep[i] != 0;
  u32 s = 0;
  u64 mykey = 0;
  if (live) {
    float4 p = newp[i];
    const float gx = p.x * nl.inv_cf, gy = p.y * nl.inv_cf, gz = p.z * nl.inv_cf;
    int ix = (int)floorf(gx), iy = (int)floorf(gy), iz = (int)floorf(gz);
    live = nl_member(nl.pruned, gx, gy, gz, ix, iy, iz, cidx % 3 - 1, (cidx / 3) % 3 - 1, cidx / 9 - 1);
    u64 key = cell_key(ix + cidx % 3 - 1, iy + (cidx / 3) % 3 - 1, iz + cidx / 9 - 1);
    mykey = key;
    s = hash_key(key) & nl.tmask;
    int probes = 0;
    while (live) {
      const u64 k = nl.table[s].key;
      if (k == key) break;
      if (k == EMPTY_KEY || ++probes > NL_MAX_PROBES) {  // (1) already reported the overflow
        live = false;
        break;
      }
      s = (s + 1) & nl.tmask;
    }
  }
  if (mode == 0) {
    if (live) atomicAdd(&nl.inc[s], 1u);
    return;
  }
  // exactly one lane per touched cell sees the batch's total for that cell (and clears it)
  u32 cnt = 0, rawcnt = 0, newcnt = 0, old = 0, st = 0, newcap = 0;
  bool mv = false, owner = false;
  if (live) {
    const u32 need = atomicExch(&nl.inc[s], 0u);
    if (need != 0) {
      rawcnt = nl.table[s].count;  // (the batch's entries are appended by the next kernel: the length and the flag BEFORE it)
      cnt = rawcnt & NL_COUNT;
      newcnt = cnt + need;
      owner = true;  // exactly one lane per touched list is here
      if (cnt + need > nl.cap[s]) {
        // new cell, or a list that outgrew its slack (the map frontier): move it to the tail with fresh slack; the
        // old storage is reclaimed by the next full rebuild
        const u32 total = cnt + need;
        newcap = nl_capacity(total);
        st = atomicAdd(&nl.state[0], newcap);
        if (st + newcap > nl.bump_end || st + newcap < st) {
          atomicExch(&nl.state[1], 1u);
          owner = false;  // (no room: the list keeps its length, the host rebuilds - nothing for k_nl_sort here)
        } else {
          mv = true;
          old = nl.table[s].start;
        }
      }
    }
  }
  const int lane = threadIdx.x & 63;
  // The touched lists of the sorted level go on the batch's work list (k_nl_sort puts them in order again). ONE global atomic
  // per workgroup: the slots are counted per wave (ballot), summed in LDS, reserved by thread 0 - a returning atomic per
  // wave on the one counter (~1 000 of them for a scan's batch) made this kernel 17 us instead of 9.
  __shared__ u32 s_wcnt[BLK / 64], s_wbase;
  {
    const unsigned long long ow = __ballot(owner && nl.sorted);
    const int wv = (int)(threadIdx.x >> 6);
    if (lane == 0) s_wcnt[wv] = (u32)__popcll(ow);
    __syncthreads();
    if (threadIdx.x == 0) {
      u32 tot = 0;
#pragma unroll
      for (int w = 0; w < BLK / 64; w++) tot += s_wcnt[w];
      s_wbase = tot ? atomicAdd(&nl.state[3], tot) : 0u;
    }
    __syncthreads();
    if (owner && nl.sorted) {
      u32 off = s_wbase;
      for (int w = 0; w < wv; w++) off += s_wcnt[w];
      // (everything k_nl_sort needs of the list: it then reads the item and the entries - no trip to the directory)
      const u32 at = off + (u32)__popcll(ow & ((1ull << lane) - 1ull));
      if (at < nl.work_cap) {
        uint4 *w = (uint4 *)nl.work + 2 * (size_t)at;
        w[0] = make_uint4(s, rawcnt, newcnt, mv ? st : nl.table[s].start);
        w[1] = make_uint4((u32)mykey, (u32)(mykey >> 32), 0u, 0u);
      }
    }
  }
  // the lists that move are copied by the whole wave, one after the other (a level-2 list has hundreds of entries)
  unsigned long long todo = __ballot(mv);
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const u32 o = __shfl(old, src), d = __shfl(st, src), n = __shfl(cnt, src);
    for (u32 j = lane; j < n; j += 64) nl.pts[(size_t)d + j] = nl.pts[(size_t)o + j];
  }
  if (mv) {
    nl.table[s].start = st;
    nl.cap[s] = newcap;
  }
}
// (2) append every kept new point (map index og_base + rank) to its 27 lists
__global__ void __launch_bounds__(BLK) k_nl_append(const float4 *__restrict__ newp, const u32 *__restrict__ keep,
                                                   const u32 *__restrict__ rank, u32 og_base, int m, NlDev nl_a,
                                                   NlDev nl_b) {
  const NlDev nl = blockIdx.y ? nl_b : nl_a;
  const long long t = (long long)blockIdx.x * BLK + threadIdx.x;
  const int i = (int)(t >> 5), cidx = (int)(t & 31);
  if (i >= m || cidx >= 27 || !keep[i]) return;
  float4 p = newp[i];
  float4 rec = make_float4(p.x, p.y, p.z, __uint_as_float(og_base + rank[i]));
  const float gx = p.x * nl.inv_cf, gy = p.y * nl.inv_cf, gz = p.z * nl.inv_cf;
  int ix = (int)floorf(gx), iy = (int)floorf(gy), iz = (int)floorf(gz);
  if (!nl_member(nl.pruned, gx, gy, gz, ix, iy, iz, cidx % 3 - 1, (cidx / 3) % 3 - 1, cidx / 9 - 1)) return;
  u64 key = cell_key(ix + cidx % 3 - 1, iy + (cidx / 3) % 3 - 1, iz + cidx / 9 - 1);
  u32 s = hash_key(key) & nl.tmask;
  int probes = 0;
  while (true) {
    const u64 k = nl.table[s].key;
    if (k == key) break;
    if (k == EMPTY_KEY || ++probes > NL_MAX_PROBES) {  // k_nl_ensure could not create the cell
      atomicExch(&nl.state[1], 1u);
      return;
    }
    s = (s + 1) & nl.tmask;
  }
  u32 pos = atomicAdd(&nl.table[s].count, 1u);
  if (pos & NL_SORTED) atomicAnd(&nl.table[s].count, NL_COUNT), pos &= NL_COUNT;  // no longer in order (k_nl_sort follows)
  if (pos >= nl.cap[s]) {  // list full: undo, the host rebuilds the lists from the map array
    atomicSub(&nl.table[s].count, 1u);
    atomicExch(&nl.state[1], 1u);
  } else {
    nl.pts[(size_t)nl.table[s].start + pos] = rec;
  }
}
// (3) a deleted map point leaves its 27 lists: the entry stays but can never be a neighbour again (x = +inf makes
//     every distance +inf, which the search drops); tombstones are swept by the next full rebuild
__global__ void __launch_bounds__(BLK) k_nl_tombstone(const float4 *__restrict__ mapp, const u32 *__restrict__ dlist,
                                                      int ndel, NlDev nl_a, NlDev nl_b) {
  const NlDev nl = blockIdx.y ? nl_b : nl_a;
  // 16 lanes per (deleted point, one of its 27 lists) on level 1 (~45 entries), a whole wave on level 2 (180..900
  // entries: 4 round trips instead of 14; the kernel is as long as its longest walk; 8 loads in flight: no shorter, round 6)
  const long long t = (long long)blockIdx.x * BLK + threadIdx.x;
  const int lg = blockIdx.y ? 6 : 4;
  const u32 lanes = 1u << lg;
  const int sub = (int)(t & (lanes - 1));
  const long long pc = t >> lg;
  const int d = (int)(pc / 27), cidx = (int)(pc % 27);
  if (d >= ndel) return;
  const u32 i = dlist[d];
  float4 p = mapp[i];
  const float gx = p.x * nl.inv_cf, gy = p.y * nl.inv_cf, gz = p.z * nl.inv_cf;
  int ix = (int)floorf(gx), iy = (int)floorf(gy), iz = (int)floorf(gz);
  // (16 lanes share (d, cidx): the whole group leaves together, the ballot below stays among lanes that walk a list)
  if (!nl_member(nl.pruned, gx, gy, gz, ix, iy, iz, cidx % 3 - 1, (cidx / 3) % 3 - 1, cidx / 9 - 1)) return;
  u64 key = cell_key(ix + cidx % 3 - 1, iy + (cidx / 3) % 3 - 1, iz + cidx / 9 - 1);
  u32 s = hash_key(key) & nl.tmask;
  while (true) {
    u64 k = nl.table[s].key;
    if (k == key) break;
    if (k == EMPTY_KEY) return;  // cannot happen for a point that was inserted
    s = (s + 1) & nl.tmask;
  }
  const u32 st = nl.table[s].start, cn = nl.table[s].count & NL_COUNT;
  const int gsh = (threadIdx.x & 63) & ~(int)(lanes - 1);  // first lane of this group inside the wave
  const unsigned long long gmask = lanes == 64 ? ~0ull : 0xFFFFull;
  for (u32 j = (u32)sub; j < cn; j += lanes * 4) {  // 4 independent loads in flight per lane
    u32 og[4];
#pragma unroll
    for (int u = 0; u < 4; u++) og[u] = __float_as_uint(nl.pts[(size_t)st + min(j + lanes * u, cn - 1)].w);
    bool hit = false;
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (og[u] == i && j + lanes * u < cn) nl.pts[(size_t)st + j + lanes * u].x = INFINITY, hit = true;
    // one entry per list matches: once a lane of the group has found it the rest of the list need not be read
    if ((__ballot(hit) >> gsh) & gmask) break;
  }
}

NlDev nl_dev(const NList &nl) {
  NlDev v;
  v.table = nl.table, v.tmask = nl.tmask, v.pts = nl.pts, v.cap = nl.cap, v.inc = nl.inc, v.state = nl.state, v.work = nl.work;
  v.work_cap = (u32)nl.work_cap;
  v.bump_end = (u32)nl.cap_pts, v.inv_cf = nl.inv_cf, v.cf = nl.cf, v.pruned = nl.pruned ? 1 : 0, v.sorted = nl.sorted ? 1 : 0;
  return v;
}


// diagnostics (malio_debug_list_order): one thread per directory slot - out[0] lists, [1] lists flagged NL_SORTED, [2] flagged
// lists whose live entries are NOT in order of distance from the cell centre, [3] live entries
__global__ void __launch_bounds__(BLK) k_nl_check(NlDev nl, unsigned long long *out) {
  const u32 s = blockIdx.x * BLK + threadIdx.x;
  if (s > nl.tmask) return;
  const Cell c = nl.table[s];
  if (c.key == EMPTY_KEY) return;
  const u32 n = c.count & NL_COUNT;
  const u64 B = 1ull << 20;
  const float cx = ((float)((int)(c.key & 0x1FFFFF) - (int)B) + 0.5f) * nl.cf,
              cy = ((float)((int)((c.key >> 21) & 0x1FFFFF) - (int)B) + 0.5f) * nl.cf,
              cz = ((float)((int)((c.key >> 42) & 0x1FFFFF) - (int)B) + 0.5f) * nl.cf;
  float last = 0.f;
  u32 live = 0;
  bool bad = false;
  for (u32 j = 0; j < n; j++) {
    const float d = nl_centre_d2(nl.pts[(size_t)c.start + j], cx, cy, cz);
    if (!(d < INFINITY)) continue;  // a tombstone
    live++;
    if (d < last) bad = true;
    last = d;
  }
  atomicAdd(&out[0], 1ull);
  if (c.count & NL_SORTED) atomicAdd(&out[1], 1ull);
  if ((c.count & NL_SORTED) && bad) atomicAdd(&out[2], 1ull);
  atomicAdd(&out[3], (unsigned long long)live);
}
int nl_check_order(Ctx *c, NList &nl, long long out4[4]) {
  unsigned long long *d = nullptr;
  MALIO_HIP(hipMalloc(&d, sizeof(unsigned long long) * 4));
  MALIO_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long) * 4, c->stream));
  hipLaunchKernelGGL(k_nl_check, dim3((nl.tmask + 1 + BLK - 1) / BLK), dim3(BLK), 0, c->stream, nl_dev(nl), d);
  hipError_t e = hipMemcpyAsync(out4, d, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  MALIO_HIP(e);
  return MALIO_OK;
}

void nl_ensure(Ctx *c, hipStream_t st, NList &nl_a, NList &nl_b, const float4 *d_new, const u32 *keep, int m,
               const MapSide &side) {
  const long long th = (long long)m * 32;
  const dim3 grid((unsigned)((th + BLK - 1) / BLK), 2);
  if (nl_a.sorted) {
    // the batch's work list (k_nl_place -> k_nl_sort): at most one item per (point, cell) pair; grown here, rarely (hipFree waits
    // for whatever still reads the old one). A list that finds no room on it stays unflagged - walked whole - until a rebuild.
    const size_t want = (size_t)std::min<long long>((long long)m * 27, 4ll << 20);
    if (want > nl_a.work_cap) {
      if (nl_a.work) (void)hipFree(nl_a.work);
      nl_a.work = nullptr, nl_a.work_cap = 0;
      const size_t cap = std::max<size_t>(want + want / 2, (size_t)1 << 16);
      if (hipMalloc(&nl_a.work, sizeof(u32) * 8 * cap) == hipSuccess) nl_a.work_cap = cap;
      else (void)hipGetLastError();
    }
  }
  const NlDev a = nl_dev(nl_a), b = nl_dev(nl_b);
  hipLaunchKernelGGL(k_nl_ensure, dim3(grid.x, 3), dim3(BLK), 0, st, d_new, keep, m, a, b, side);
  hipLaunchKernelGGL(k_nl_place, grid, dim3(BLK), 0, st, d_new, keep, m, a, b, 1);
}
void nl_append(Ctx *c, hipStream_t st, NList &nl_a, NList &nl_b, const float4 *d_new, const u32 *keep, const u32 *rank, u32 og_base,
               int m) {
  const long long th = (long long)m * 32;
  hipLaunchKernelGGL(k_nl_append, dim3((unsigned)((th + BLK - 1) / BLK), 2), dim3(BLK), 0, st, d_new, keep, rank,
                     og_base, m, nl_dev(nl_a), nl_dev(nl_b));
  // the lists of the sorted level this batch appended to, in order again (one wave per list; the work list's length is on the
  // device: a grid that covers a typical batch in one round, the rest by striding)
  if (nl_a.sorted) {  // (no work list - its allocation failed -: the kernel sweeps the directory)
    const NlDev a = nl_dev(nl_a);
    const long long lists = std::min<long long>((long long)m * 27, (long long)nl_a.tmask + 1);
    // (grid: a typical batch - 1 600 new points, ~8 k lists - in two rounds of one list per wave)
    hipLaunchKernelGGL(k_nl_sort, dim3((unsigned)std::max<long long>(1, std::min<long long>(2048, (lists * 64 + BLK - 1) / BLK))),
                       dim3(BLK), 0, st, a, (const u32 *)a.work, (const u32 *)(a.state + 3));
  }
}
void nl_tombstone(Ctx *c, hipStream_t st, NList &nl_a, NList &nl_b, const float4 *d_map, const u32 *dlist, int ndel) {
  const long long th = (long long)ndel * 27 * 64;  // (level 2's need; the level-1 half of the grid leaves early)
  hipLaunchKernelGGL(k_nl_tombstone, dim3((unsigned)((th + BLK - 1) / BLK), 2), dim3(BLK), 0, st, d_map, dlist,
                     ndel, nl_dev(nl_a), nl_dev(nl_b));
}

}  // namespace malio
