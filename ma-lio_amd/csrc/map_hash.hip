// GPU-resident spatial hash that replaces the ikd-Tree as the 5-NN search structure
// (reference: KD_TREE<PointType>, include/ikd-Tree/ikd_Tree.{h,cpp}; Build :369-397).
// Layout in HBM: points sorted by cell as float4 (x,y,z,bits(original index)) + the original-order
// float4 (x,y,z,normal_y) array the plane fit gathers from + a compact open-addressing table
// of 16-byte {key,start,count} entries. Cell edge c >= sqrt(5) m so the 27 cells around a query
// contain every map point within the reference's acceptance radius (laserMapping.cpp:587).
#include "malio_internal.hpp"

namespace malio {

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

// Pass 1: insert every point's cell key into a big scratch table, take a rank inside the cell.
__global__ void __launch_bounds__(BLK) k_gbc_insert(const float4 *__restrict__ pts, int n, float inv_c, u64 *keys,
                                                    u32 *cnt, u32 mask, u32 *slot_of, u32 *rank_of, u32 *ncells) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  int ix = (int)floorf(p.x * inv_c), iy = (int)floorf(p.y * inv_c), iz = (int)floorf(p.z * inv_c);
  u64 key = cell_key(ix, iy, iz);
  u32 s = hash_key(key) & mask;
  while (true) {
    u64 old = atomicCAS(&keys[s], EMPTY_KEY, key);
    if (old == EMPTY_KEY) {
      atomicAdd(ncells, 1u);
      break;
    }
    if (old == key) break;
    s = (s + 1) & mask;
  }
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

__device__ __forceinline__ u32 brick_hash(int bx, int by, int bz) {  // == brick_hash_d() in measure.hip
  u32 h = (u32)bx * 0x9E3779B1u ^ (u32)by * 0x85EBCA77u ^ (u32)bz * 0xC2B2AE3Du;
  h ^= h >> 15;
  h *= 0x27D4EB2Fu;
  h ^= h >> 13;
  return h;
}
__global__ void __launch_bounds__(BLK) k_gbc_occ(const Cell *__restrict__ table, u32 tsize, u64 *occ, u32 omask) {
  u32 s = blockIdx.x * BLK + threadIdx.x;
  if (s >= tsize) return;
  Cell c = table[s];
  if (c.key == EMPTY_KEY || c.count == 0) return;
  const long long B = 1ll << 20;
  int ix = (int)((long long)(c.key & 0x1FFFFF) - B);
  int iy = (int)((long long)((c.key >> 21) & 0x1FFFFF) - B);
  int iz = (int)((long long)((c.key >> 42) & 0x1FFFFF) - B);
  u32 line = brick_hash(ix >> 3, iy >> 3, iz >> 3) & omask;
  atomicOr(&occ[(size_t)line * 8 + (iz & 7)], 1ull << ((ix & 7) + 8 * (iy & 7)));
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

static u32 next_pow2(u32 v) {
  u32 p = 1;
  while (p < v) p <<= 1;
  return p;
}

void free_grid(CellGrid &g) {
  if (g.table) (void)hipFree(g.table);
  if (g.occ) (void)hipFree(g.occ);
  if (g.pts) (void)hipFree(g.pts);
  if (g.orig) (void)hipFree(g.orig);
  g = CellGrid();
}

static int exclusive_scan_u32(Ctx *c, const u32 *d_in, u32 *d_out, u32 *d_tiles, int n) {
  int ntiles = (n + 1023) / 1024;
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(BLK), 0, c->stream, d_in, d_out, d_tiles, n);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(BLK), 0, c->stream, d_tiles, ntiles);
  hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(BLK), 0, c->stream, d_out, d_tiles, n);
  return MALIO_OK;
}

int group_by_cell(Ctx *c, const float4 *d_in, int n, float inv_cell, CellGrid &g, const u32 *d_in_orig) {
  if (n <= 0) {
    g.n = 0;
    return MALIO_OK;
  }
  u32 tbig = next_pow2((u32)std::max(1024, 2 * n));
  u64 *keys = nullptr;
  u32 *cnt = nullptr, *start = nullptr, *slot_of = nullptr, *rank_of = nullptr, *tiles = nullptr, *ncells = nullptr;
  int ntiles = (tbig + 1023) / 1024;
  MALIO_HIP(hipMalloc(&keys, sizeof(u64) * tbig));
  MALIO_HIP(hipMalloc(&cnt, sizeof(u32) * tbig));
  MALIO_HIP(hipMalloc(&start, sizeof(u32) * tbig));
  MALIO_HIP(hipMalloc(&slot_of, sizeof(u32) * n));
  MALIO_HIP(hipMalloc(&rank_of, sizeof(u32) * n));
  MALIO_HIP(hipMalloc(&tiles, sizeof(u32) * (ntiles + 1)));
  MALIO_HIP(hipMalloc(&ncells, sizeof(u32)));
  hipLaunchKernelGGL(k_fill_u64, dim3((tbig + BLK - 1) / BLK), dim3(BLK), 0, c->stream, keys, EMPTY_KEY, (size_t)tbig);
  MALIO_HIP(hipMemsetAsync(cnt, 0, sizeof(u32) * tbig, c->stream));
  MALIO_HIP(hipMemsetAsync(ncells, 0, sizeof(u32), c->stream));
  int nb = (n + BLK - 1) / BLK;
  hipLaunchKernelGGL(k_gbc_insert, dim3(nb), dim3(BLK), 0, c->stream, d_in, n, inv_cell, keys, cnt, tbig - 1, slot_of,
                     rank_of, ncells);
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
  u32 h_ncells = 0;
  MALIO_HIP(hipMemcpyAsync(&h_ncells, ncells, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  u32 tsize = next_pow2(std::max(1024u, 4u * h_ncells));
  if ((size_t)tsize > g.cap_table) {
    if (g.table) (void)hipFree(g.table);
    g.table = nullptr;
    g.cap_table = tsize;
    MALIO_HIP(hipMalloc(&g.table, sizeof(Cell) * g.cap_table));
  }
  hipLaunchKernelGGL(k_clear_table, dim3((tsize + BLK - 1) / BLK), dim3(BLK), 0, c->stream, g.table, tsize);
  hipLaunchKernelGGL(k_gbc_compact, dim3((tbig + BLK - 1) / BLK), dim3(BLK), 0, c->stream, keys, cnt, start, tbig,
                     g.table, tsize - 1);
  // occupancy filter: >= 8 lines per occupied brick-equivalent keeps false positives rare; 2 MB at 1M points
  u32 olines = next_pow2(std::max(1024u, h_ncells / 4));
  if ((size_t)olines * 8 > g.cap_occ) {
    if (g.occ) (void)hipFree(g.occ);
    g.occ = nullptr;
    g.cap_occ = (size_t)olines * 8;
    MALIO_HIP(hipMalloc(&g.occ, sizeof(u64) * g.cap_occ));
  }
  MALIO_HIP(hipMemsetAsync(g.occ, 0, sizeof(u64) * (size_t)olines * 8, c->stream));
  hipLaunchKernelGGL(k_gbc_occ, dim3((tsize + BLK - 1) / BLK), dim3(BLK), 0, c->stream, g.table, tsize, g.occ,
                     olines - 1);
  g.omask = olines - 1;
  MALIO_HIP(hipStreamSynchronize(c->stream));
  g.tmask = tsize - 1;
  g.ncells = h_ncells;
  g.n = n;
  (void)hipFree(keys);
  (void)hipFree(cnt);
  (void)hipFree(start);
  (void)hipFree(slot_of);
  (void)hipFree(rank_of);
  (void)hipFree(tiles);
  (void)hipFree(ncells);
  MALIO_HIP(hipGetLastError());
  return MALIO_OK;
}

}  // namespace malio
