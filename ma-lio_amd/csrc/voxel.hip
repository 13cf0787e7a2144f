// Voxel down-sampling of an undistorted cloud: what the reference does between the
// undistortion and the measurement update with `downSizeFilterSurf.setInputCloud(...); .filter(...)`
// (src/laserMapping.cpp:93,860,968-971) - pcl::VoxelGrid<pcl::PointXYZINormal> with its defaults
// (downsample_all_data = true, min_points_per_voxel = 0, no filter field), one call per LiDAR.
//
// PCL is a third-party dependency that is NOT under /root/reference (CMakeLists.txt:56 `find_package(PCL 1.8
// REQUIRED)`, unpinned; Ubuntu 20.04 ships 1.10), so this restates its published algorithm
// (filters/impl/voxel_grid.hpp `applyFilter`, common/impl/accumulators.hpp) - PARITY IS UNPINNED:
//   1. bounding box of the finite points; inverse leaf = 1 / leaf (float)
//   2. min_b = floor(min * inv), max_b = floor(max * inv), div_b = max_b - min_b + 1; if dx*dy*dz exceeds INT_MAX
//      the filter warns and returns the input unchanged
//   3. idx = (floor(x*inv) - min_b.x) + (floor(y*inv) - min_b.y) * div.x + (floor(z*inv) - min_b.z) * div.x*div.y
//   4. points sorted by idx; one output point per distinct idx, in ascending idx order
//   5. centroid of every field: xyz, intensity, curvature = float sum / n; the normal (normal_x/y/z + pad) is
//      summed and then either divided by n or normalised to unit length - PCL changed this between releases
//      (CentroidPoint's AccumulatorNormal normalises), hence the `normal_mode` switch. The mapping loop overwrites
//      normal_x right after the filter (:973) and the update rewrites normal_y, so the choice rarely matters.
// Within a voxel PCL adds the points in whatever order its (unstable) integer sort left them; here the sort is
// stable, i.e. input order - float sums can differ from a PCL build in the last bits for that reason alone.
#include "malio_internal.hpp"

namespace malio {
namespace {

__device__ __forceinline__ u32 fenc(float x) {  // order-preserving float -> u32
  u32 b = __float_as_uint(x);
  return (b >> 31) ? ~b : (b | 0x80000000u);
}
inline float fdec_host(u32 k) {
  u32 b = (k >> 31) ? (k & 0x7FFFFFFFu) : ~k;
  float f;
  memcpy(&f, &b, 4);
  return f;
}

constexpr int VG_SLOTS = 64;
// 12 floats per point, pcl::PointXYZINormal layout: x y z _ nx ny nz _ intensity curvature _ _
__global__ void __launch_bounds__(BLK) k_vg_bounds(const float *__restrict__ pts, int n,
                                                   u32 *mm /*[VG_SLOTS][6] min xyz, max xyz*/) {
  int i = blockIdx.x * BLK + threadIdx.x;
  float v[3] = {INFINITY, INFINITY, INFINITY}, w[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (i < n) {
    const float x = pts[(size_t)i * 12], y = pts[(size_t)i * 12 + 1], z = pts[(size_t)i * 12 + 2];
    if (isfinite(x) && isfinite(y) && isfinite(z)) v[0] = w[0] = x, v[1] = w[1] = y, v[2] = w[2] = z;
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      v[a] = fminf(v[a], __shfl_xor(v[a], d));
      w[a] = fmaxf(w[a], __shfl_xor(w[a], d));
    }
  }
  // workgroup result through LDS, then ONE set of atomics per workgroup on a slot chosen by workgroup id: 3 125
  // waves hitting the same six addresses cost 215 us (same-address atomics serialise at ~12 ns), this costs 5
  __shared__ float sm[BLK / 64][6];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) sm[wave][a] = v[a], sm[wave][3 + a] = w[a];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    float mn = sm[0][a], mx = sm[0][3 + a];
#pragma unroll
    for (int k = 1; k < BLK / 64; k++) mn = fminf(mn, sm[k][a]), mx = fmaxf(mx, sm[k][3 + a]);
    u32 *slot = mm + (size_t)(blockIdx.x & (VG_SLOTS - 1)) * 6;
    if (mn != INFINITY) atomicMin(&slot[a], fenc(mn));
    if (mx != -INFINITY) atomicMax(&slot[3 + a], fenc(mx));
  }
}

struct VgGrid {
  float inv[3];
  int min_b[3];
  int mul[3];
};

__global__ void __launch_bounds__(BLK) k_vg_keys(const float *__restrict__ pts, int n, VgGrid g, u32 *keys, u32 *vals) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  const float x = pts[(size_t)i * 12], y = pts[(size_t)i * 12 + 1], z = pts[(size_t)i * 12 + 2];
  u32 key = 0xFFFFFFFFu;  // non-finite points sort to the end and are dropped (PCL skips them)
  if (isfinite(x) && isfinite(y) && isfinite(z)) {
    const int i0 = (int)(floorf(x * g.inv[0]) - (float)g.min_b[0]);
    const int i1 = (int)(floorf(y * g.inv[1]) - (float)g.min_b[1]);
    const int i2 = (int)(floorf(z * g.inv[2]) - (float)g.min_b[2]);
    key = (u32)(i0 * g.mul[0] + i1 * g.mul[1] + i2 * g.mul[2]);
  }
  keys[i] = key;
  vals[i] = (u32)i;
}

// ---- stable LSD radix sort of (voxel index, point index) pairs, 8 bits per pass ---------------------------------------------
// A wave owns a CHUNK of 1 024 consecutive pairs (all loaded up front) and walks it 64 at a time, in order: per pass one kernel counts the
// chunk's digits (LDS atomics), one scan over the [digit][chunk] table turns the counts into positions, one kernel
// places the pairs - a lane's rank among the earlier pairs of its digit = the chunk's running count of that digit (LDS)
// + the lanes below it in the wave that hold the same digit (nine ballots match the digit). Stable by construction:
// digit-major scan, chunks ascending, the walk in order. Only as many passes as the largest voxel index has bytes.
constexpr int RS_CHUNK = 1024;
constexpr int RS_IT = RS_CHUNK / 64;
template <int DB>  // digit bits
__device__ __forceinline__ unsigned long long same_digit_lanes(u32 d, bool valid) {
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < DB; b++) {
    const unsigned long long s = __ballot(valid && ((d >> b) & 1u));
    m &= ((d >> b) & 1u) ? s : ~s;
  }
  return m;  // (meaningful in the valid lanes)
}
template <int DB>
__global__ void __launch_bounds__(BLK) k_rs_hist(const u32 *__restrict__ keys, int n, int shift, u32 *hist, int nchunks) {
  constexpr int NB = 1 << DB;
  __shared__ u32 h[BLK / 64][NB];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * (BLK / 64) + wv;
  for (int d = lane; d < NB; d += 64) h[wv][d] = 0;
  if (chunk >= nchunks) return;  // (no workgroup barrier below: the waves are independent)
  u32 key[RS_IT];  // the whole chunk's loads in flight at once: the walk below is a chain of LDS operations, not of HBM trips
#pragma unroll
  for (int it = 0; it < RS_IT; it++) {
    const int i = chunk * RS_CHUNK + it * 64 + lane;
    key[it] = i < n ? keys[i] : 0u;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < RS_IT; it++)
    if (chunk * RS_CHUNK + it * 64 + lane < n) atomicAdd(&h[wv][(key[it] >> shift) & (NB - 1)], 1u);
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < NB; d += 64) hist[(size_t)d * nchunks + chunk] = h[wv][d];
}
template <int DB>
__global__ void __launch_bounds__(BLK) k_rs_scatter(const u32 *__restrict__ kin, const u32 *__restrict__ vin, u32 *kout, u32 *vout,
                                                    int n, int shift, const u32 *__restrict__ offs, int nchunks) {
  constexpr int NB = 1 << DB;
  __shared__ u32 run[BLK / 64][NB];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * (BLK / 64) + wv;
  if (chunk >= nchunks) return;  // (no workgroup barrier below: the waves are independent)
  u32 key[RS_IT], val[RS_IT];
#pragma unroll
  for (int it = 0; it < RS_IT; it++) {
    const int i = chunk * RS_CHUNK + it * 64 + lane;
    key[it] = i < n ? kin[i] : 0u, val[it] = i < n ? vin[i] : 0u;
  }
  for (int d = lane; d < NB; d += 64) run[wv][d] = offs[(size_t)d * nchunks + chunk];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < RS_IT; it++) {
    const bool valid = chunk * RS_CHUNK + it * 64 + lane < n;
    const u32 d = (key[it] >> shift) & (NB - 1);
    const unsigned long long m = same_digit_lanes<DB>(d, valid);
    const u32 below = (u32)__popcll(m & ((1ull << lane) - 1ull));
    u32 base = 0;
    if (valid) base = run[wv][d];
    __builtin_amdgcn_wave_barrier();  // every lane has read its digit's count before a leader moves it on
    if (valid && below == 0) run[wv][d] = base + (u32)__popcll(m);
    __builtin_amdgcn_wave_barrier();
    if (valid) kout[base + below] = key[it], vout[base + below] = val[it];
  }
}
// sorts (k1, v1) by the low `bits` bits of the key (9-bit digits: three passes up to 2^27 voxels, four beyond); the result
// is in (k1, v1) again when it returns
static int radix_sort_pairs(Ctx *c, ArenaScope &sc, u32 *&k1, u32 *&k2, u32 *&v1, u32 *&v2, int n, int bits) {
  constexpr int DB = 9, NB = 1 << DB;
  const int nchunks = (n + RS_CHUNK - 1) / RS_CHUNK;
  const int nh = NB * nchunks;
  u32 *hist = nullptr, *offs = nullptr, *tiles = nullptr;
  MALIO_HIP(sc.get(&hist, (size_t)nh));
  MALIO_HIP(sc.get(&offs, (size_t)nh));
  MALIO_HIP(sc.get(&tiles, (size_t)(nh + 1023) / 1024 + 2));
  const dim3 grid((nchunks + BLK / 64 - 1) / (BLK / 64));
  for (int shift = 0; shift < bits; shift += DB) {
    hipLaunchKernelGGL(k_rs_hist<DB>, grid, dim3(BLK), 0, c->stream, k1, n, shift, hist, nchunks);
    if (int rc = exclusive_scan_u32(c, hist, offs, tiles, nh)) return rc;
    hipLaunchKernelGGL(k_rs_scatter<DB>, grid, dim3(BLK), 0, c->stream, k1, v1, k2, v2, n, shift, offs, nchunks);
    std::swap(k1, k2), std::swap(v1, v2);
  }
  MALIO_HIP(hipGetLastError());
  return MALIO_OK;
}

__global__ void __launch_bounds__(BLK) k_vg_heads(const u32 *__restrict__ keys, int n, u32 *head) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i > n) return;  // head[n] = 0 so that the exclusive scan leaves the total at [n]
  head[i] = (i < n && keys[i] != 0xFFFFFFFFu && (i == 0 || keys[i] != keys[i - 1])) ? 1u : 0u;
}
__global__ void __launch_bounds__(BLK) k_vg_first(const u32 *__restrict__ head, const u32 *__restrict__ pos, int n,
                                                  u32 *first) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i < n && head[i]) first[pos[i]] = (u32)i;
}

// one thread per output voxel: float sums in sorted (= input) order, then the accumulators' get()
__global__ void __launch_bounds__(BLK) k_vg_centroid(const float *__restrict__ pts, const u32 *__restrict__ keys,
                                                     const u32 *__restrict__ vals, const u32 *__restrict__ first, int nvox,
                                                     int n, int normal_mode, float *out) {
  int o = blockIdx.x * BLK + threadIdx.x;
  if (o >= nvox) return;
  const u32 b = first[o];
  const u32 key = keys[b];
  float sx = 0, sy = 0, sz = 0, nx = 0, ny = 0, nz = 0, nw = 0, si = 0, sc = 0;
  u32 cnt = 0;
  for (u32 j = b; j < (u32)n && keys[j] == key; j++, cnt++) {
    const float *p = pts + (size_t)vals[j] * 12;
    sx += p[0], sy += p[1], sz += p[2];
    nx += p[4], ny += p[5], nz += p[6], nw += p[7];
    si += p[8], sc += p[9];
  }
  const float fn = (float)cnt;
  float *q = out + (size_t)o * 12;
  q[0] = sx / fn, q[1] = sy / fn, q[2] = sz / fn, q[3] = 1.0f;
  if (normal_mode == MALIO_VOXEL_NORMAL_NORMALIZE) {
    const float len = sqrtf(nx * nx + ny * ny + nz * nz + nw * nw);  // Eigen: v /= v.norm(), zero stays zero
    if (len > 0.f) nx /= len, ny /= len, nz /= len, nw /= len;
    q[4] = nx, q[5] = ny, q[6] = nz, q[7] = nw;
  } else {
    q[4] = nx / fn, q[5] = ny / fn, q[6] = nz / fn, q[7] = nw / fn;
  }
  q[8] = si / fn, q[9] = sc / fn, q[10] = 0.f, q[11] = 0.f;
}

// extrema slots: [3 min | 3 max] as order-preserving u32
__global__ void k_vg_init(u32 *mm) { mm[threadIdx.x] = (threadIdx.x % 6) < 3 ? 0xFFFFFFFFu : 0u; }

}  // namespace

// (map_update.hip: the map array's cell order) sorts (k1, v1) by the low `bits` bits of the key, stable; scratch from `sc`
int radix_sort_pairs_u32(Ctx *c, ArenaScope &sc, u32 *&k1, u32 *&k2, u32 *&v1, u32 *&v2, int n, int bits) {
  return radix_sort_pairs(c, sc, k1, k2, v1, v2, n, bits);
}

// Device core. d_pts: [n][12] in HBM. On return *d_out (arena memory of the CALLER's scope: `sc`) holds *out_n points,
// or *passthrough is set when PCL's "leaf size too small" branch applies (output = input).
int voxel_downsample_dev(Ctx *c, ArenaScope &sc, const float *d_pts, int n, float leaf, int normal_mode, float **d_out,
                         int *out_n, bool *passthrough) {
  *out_n = 0, *d_out = nullptr, *passthrough = false;
  if (n <= 0) return MALIO_OK;
  u32 *d_mm = nullptr, *k1 = nullptr, *k2 = nullptr, *v1 = nullptr, *v2 = nullptr, *head = nullptr, *pos = nullptr,
      *first = nullptr, *tiles = nullptr;
  MALIO_HIP(sc.get(&d_mm, (size_t)VG_SLOTS * 6));
  u32 *mb = nullptr, *mbd = nullptr;
  MALIO_HIP(mbox(c, &mb, &mbd));
  u32 *mms = mb + 1024;  // pinned: [VG_SLOTS][3 min | 3 max]
  static_assert(1024 + VG_SLOTS * 6 + 1 <= MBOX_WORDS, "mailbox too small");
  hipLaunchKernelGGL(k_vg_init, dim3(1), dim3(VG_SLOTS * 6), 0, c->stream, d_mm);
  const int nb = (n + BLK - 1) / BLK;
  hipLaunchKernelGGL(k_vg_bounds, dim3(nb), dim3(BLK), 0, c->stream, d_pts, n, d_mm);
  MALIO_HIP(hipMemcpyAsync(mms, d_mm, sizeof(u32) * VG_SLOTS * 6, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  u32 mm[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  for (int s = 0; s < VG_SLOTS; s++)
    for (int a = 0; a < 3; a++) mm[a] = std::min(mm[a], mms[s * 6 + a]), mm[3 + a] = std::max(mm[3 + a], mms[s * 6 + 3 + a]);
  if (mm[0] == 0xFFFFFFFFu) return MALIO_OK;  // no finite point at all
  VgGrid g;
  int div[3];
  // PCL forms dx*dy*dz in int64 and compares with INT_MAX; that product itself wraps for absurd extents, so the test
  // is applied after every factor (same answer whenever PCL's arithmetic is defined)
  bool too_small = false;
  long long cells = 1;
  for (int a = 0; a < 3; a++) {
    const float mn = fdec_host(mm[a]), mx = fdec_host(mm[3 + a]);
    g.inv[a] = 1.0f / leaf;
    g.min_b[a] = (int)std::floor(mn * g.inv[a]);
    div[a] = (int)std::floor(mx * g.inv[a]) - g.min_b[a] + 1;
    const float ext = (mx - mn) * g.inv[a];
    if (!(ext < 4.0e9f)) too_small = true;
    if (!too_small) {
      cells *= (long long)ext + 1;
      if (cells > 0x7FFFFFFFll) too_small = true;
    }
  }
  if (too_small) {  // "Leaf size is too small for the input dataset. Integer indices would overflow." -> output = input
    *passthrough = true;
    *out_n = n;
    return MALIO_OK;
  }
  g.mul[0] = 1, g.mul[1] = div[0], g.mul[2] = div[0] * div[1];
  const long long cells_total = (long long)div[0] * div[1] * div[2];  // (<= INT_MAX: checked above on the float extents;
                                                                        //  the int form can exceed the float form by one cell per axis)
  MALIO_HIP(sc.get(&k1, (size_t)n));
  MALIO_HIP(sc.get(&k2, (size_t)n));
  MALIO_HIP(sc.get(&v1, (size_t)n));
  MALIO_HIP(sc.get(&v2, (size_t)n));
  MALIO_HIP(sc.get(&head, (size_t)n + 1));
  MALIO_HIP(sc.get(&pos, (size_t)n + 1));
  MALIO_HIP(sc.get(&first, (size_t)n + 1));
  MALIO_HIP(sc.get(&tiles, (size_t)(n + 1 + 1023) / 1024 + 2));
  hipLaunchKernelGGL(k_vg_keys, dim3(nb), dim3(BLK), 0, c->stream, d_pts, n, g, k1, v1);
  {
    // valid keys are < cells <= INT_MAX; the key of a non-finite point is all ones: its low bits sort it behind every
    // valid key as long as 2^bits > cells
    int bits = 1;
    while (bits < 32 && (1ll << bits) <= cells_total) bits++;
    if (int rcs = radix_sort_pairs(c, sc, k1, k2, v1, v2, n, bits)) return rcs;
    std::swap(k1, k2), std::swap(v1, v2);  // (the code below reads the sorted pairs from k2 / v2)
  }
  hipLaunchKernelGGL(k_vg_heads, dim3((n + 1 + BLK - 1) / BLK), dim3(BLK), 0, c->stream, k2, n, head);
  u32 *h_nvox = mb + 1024 + VG_SLOTS * 6;
  exclusive_scan_u32(c, head, pos, tiles, n + 1, mbd + 1024 + VG_SLOTS * 6);  // the voxel count lands in the mapped buffer
  hipLaunchKernelGGL(k_vg_first, dim3(nb), dim3(BLK), 0, c->stream, head, pos, n, first);
  MALIO_HIP(hipStreamSynchronize(c->stream));
  const u32 nvox = *h_nvox;
  *out_n = (int)nvox;
  if (nvox == 0) return MALIO_OK;
  MALIO_HIP(sc.get(d_out, (size_t)nvox * 12));
  hipLaunchKernelGGL(k_vg_centroid, dim3((nvox + BLK - 1) / BLK), dim3(BLK), 0, c->stream, d_pts, k2, v2, first, (int)nvox,
                     n, normal_mode, *d_out);
  MALIO_HIP(hipGetLastError());
  return MALIO_OK;
}

int voxel_downsample(Ctx *c, const malio_point_t *pts, int n, float leaf, int normal_mode, malio_point_t *out, int cap,
                     int *out_n) {
  MALIO_HIP(hipSetDevice(c->device));
  *out_n = 0;
  if (n <= 0) return MALIO_OK;
  if (!(leaf > 0.f)) {
    c->err = "malio_voxel_downsample: leaf size must be positive";
    return MALIO_ERR_BAD_ARG;
  }
  ArenaScope sc(c->arena);
  float *d_pts = nullptr, *d_out = nullptr;
  MALIO_HIP(sc.get(&d_pts, (size_t)n * 12));
  MALIO_HIP(hipMemcpyAsync(d_pts, pts, sizeof(float) * 12 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  bool pass = false;
  int m = 0;
  int rc = voxel_downsample_dev(c, sc, d_pts, n, leaf, normal_mode, &d_out, &m, &pass);
  if (rc != MALIO_OK) return rc;
  *out_n = m;
  const int take = std::min(m, cap);
  if (take <= 0) return MALIO_OK;
  if (pass) {
    memcpy(out, pts, sizeof(malio_point_t) * (size_t)take);
    return MALIO_OK;
  }
  MALIO_HIP(hipMemcpyAsync(out, d_out, sizeof(float) * 12 * (size_t)take, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  return MALIO_OK;
}

}  // namespace malio
