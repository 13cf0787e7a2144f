// Sensor decode: dataset records -> the point cloud the mapping loop consumes, i.e. what
// file_player's readers plus Preprocess::process do between a .bin file of the City dataset and `lidar_buffer`:
//   Livox  (Avia / Tele): 19-byte records x y z (f32), reflectivity tag line (u8), offset_time (u32)
//                         file_player/src/ROSThread.cpp:776-796, 817-833  ->  Preprocess::avia_handler,
//                         MA_LIO/src/preprocess.cpp:59-107
//   Ouster:               22-byte records x y z intensity (f32), ring (u16), t (u32)
//                         file_player/src/ROSThread.cpp:947-957           ->  Preprocess::oust64_handler,
//                         MA_LIO/src/preprocess.cpp:109-149
//   Velodyne:             the data[] of a sensor_msgs::PointCloud2 (fields by offset, see malio_pc2_layout_t)
//                         pcl::fromROSMsg + Preprocess::velodyne_handler, MA_LIO/src/preprocess.cpp:148-212
// Byte/integer work bounded by HBM: one thread per record, unaligned little-endian loads, the handler's filters,
// then an order-preserving compaction (exclusive scan of the keep flags). The handlers' sequential pieces are
// restated in parallel form:
//   avia: `valid_num` (:85) is an inclusive prefix count of the tag/line test, a point is looked at when
//         valid_num % point_filter_num == 0 (:86); the "differs from the previous point" test (:96) reads
//         pl_full[i-1], which holds point i-1's values only if i-1 itself was looked at (pl_full is zero-filled by
//         resize, :69) - so the previous record is decoded again under that condition. The && / || precedence of
//         :96 is kept as written: dx || dy || (dz && outside-blind).
#include "malio_internal.hpp"

namespace malio {
namespace {

__device__ __forceinline__ float ld_f32(const unsigned char *p) {
  u32 v = (u32)p[0] | (u32)p[1] << 8 | (u32)p[2] << 16 | (u32)p[3] << 24;
  return __uint_as_float(v);
}
__device__ __forceinline__ u32 ld_u32(const unsigned char *p) {
  return (u32)p[0] | (u32)p[1] << 8 | (u32)p[2] << 16 | (u32)p[3] << 24;
}

struct LivoxRec {
  float x, y, z;
  unsigned char reflectivity, tag, line;
  u32 offset_time;
};
__device__ __forceinline__ LivoxRec livox_rec(const unsigned char *rec, int n_rec, int i) {
  LivoxRec r;
  if (i >= n_rec) {  // file_player's `while(!file.eof())` pushes one more, default-constructed point (:780-792)
    r.x = r.y = r.z = 0.f, r.reflectivity = r.tag = r.line = 0, r.offset_time = 0;
    return r;
  }
  const unsigned char *p = rec + (size_t)i * 19;
  r.x = ld_f32(p), r.y = ld_f32(p + 4), r.z = ld_f32(p + 8);
  r.reflectivity = p[12], r.tag = p[13], r.line = p[14];
  r.offset_time = ld_u32(p + 15);
  return r;
}
__device__ __forceinline__ bool livox_valid(const LivoxRec &r, int n_scans) {  // preprocess.cpp:82
  return ((int)r.line < n_scans) && ((r.tag & 0x30) == 0x10 || (r.tag & 0x30) == 0x00);
}

__global__ void __launch_bounds__(BLK) k_livox_valid(const unsigned char *__restrict__ rec, int n_rec, int plsize, int n_scans,
                                                     u32 *valid) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i > plsize) return;  // valid[plsize] = 0 (scan tail)
  valid[i] = (i >= 1 && i < plsize && livox_valid(livox_rec(rec, n_rec, i), n_scans)) ? 1u : 0u;  // the loop starts at 1 (:80)
}

// vnum[i] = number of valid points among 1..i-1 (exclusive scan); valid_num at i is vnum[i] + valid[i]
__global__ void __launch_bounds__(BLK) k_livox_select(const unsigned char *__restrict__ rec, int n_rec, int plsize, int n_scans,
                                                      int pfn, double blind, const u32 *__restrict__ valid,
                                                      const u32 *__restrict__ vnum, u32 *keep, u32 *tmax_bits) {
  int i = blockIdx.x * BLK + threadIdx.x;
  float cur = -INFINITY;
  bool push = false;
  if (i >= 1 && i < plsize && valid[i] && ((vnum[i] + 1u) % (u32)pfn) == 0u) {
    const LivoxRec r = livox_rec(rec, n_rec, i);
    const float curvature = (float)r.offset_time / float(1000000);  // :92, ms
    if (!(curvature > 100)) {                                         // :93-94
      cur = curvature;                                                // :95-96 (maximum_time)
      // pl_full[i-1]: the previous point's x y z if it was looked at, else the zeros of resize()
      float px = 0.f, py = 0.f, pz = 0.f;
      if (i - 1 >= 1 && valid[i - 1] && ((vnum[i - 1] + 1u) % (u32)pfn) == 0u) {
        const LivoxRec q = livox_rec(rec, n_rec, i - 1);
        px = q.x, py = q.y, pz = q.z;
      }
      const bool dx = fabsf(r.x - px) > 1e-7, dy = fabsf(r.y - py) > 1e-7, dz = fabsf(r.z - pz) > 1e-7;
      const bool far = (double)(r.x * r.x + r.y * r.y + r.z * r.z) > blind * blind;
      push = dx || dy || (dz && far);  // :96, precedence as written
    }
  }
  if (i <= plsize) keep[i] = push ? 1u : 0u;
  // maximum_time: curvatures are >= 0, so their float bits order like the values
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cur = fmaxf(cur, __shfl_xor(cur, d));
  if ((threadIdx.x & 63) == 0 && cur >= 0.f) atomicMax(&tmax_bits[blockIdx.x & 63], __float_as_uint(cur) + 1u);  // 0 = none
}

__global__ void __launch_bounds__(BLK) k_livox_emit(const unsigned char *__restrict__ rec, int n_rec, int plsize,
                                                    const u32 *__restrict__ keep, const u32 *__restrict__ pos, float *out12) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= plsize || !keep[i]) return;
  const LivoxRec r = livox_rec(rec, n_rec, i);
  float *q = out12 + (size_t)pos[i] * 12;
  q[0] = r.x, q[1] = r.y, q[2] = r.z, q[3] = 1.f;
  q[4] = 0.f, q[5] = 0.f, q[6] = 0.f, q[7] = 0.f;
  q[8] = (float)r.reflectivity;                    // :90
  q[9] = (float)r.offset_time / float(1000000);   // :92
  q[10] = 0.f, q[11] = 0.f;
}

// ---- Ouster ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLK) k_ouster_select(const unsigned char *__restrict__ rec, int n, int pfn, double blind,
                                                       float time_unit_scale, u32 *keep, u32 *tmax_bits) {
  int i = blockIdx.x * BLK + threadIdx.x;
  float cur = -INFINITY;
  bool push = false;
  if (i < n && i % pfn == 0) {  // :122-123
    const unsigned char *p = rec + (size_t)i * 22;
    const float x = ld_f32(p), y = ld_f32(p + 4), z = ld_f32(p + 8);
    const double range = (double)(x * x + y * y + z * z);  // :125 (float arithmetic, widened on assignment)
    if (!(range < blind * blind)) {                         // :127-128
      push = true;
      cur = (float)ld_u32(p + 18) * time_unit_scale * 1.e-9f;  // :139
    }
  }
  if (i <= n) keep[i] = push ? 1u : 0u;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cur = fmaxf(cur, __shfl_xor(cur, d));
  if ((threadIdx.x & 63) == 0 && cur >= 0.f) atomicMax(&tmax_bits[blockIdx.x & 63], __float_as_uint(cur) + 1u);  // 0 = none
}
__global__ void __launch_bounds__(BLK) k_ouster_emit(const unsigned char *__restrict__ rec, int n, float time_unit_scale,
                                                     const u32 *__restrict__ keep, const u32 *__restrict__ pos, float *out12) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n || !keep[i]) return;
  const unsigned char *p = rec + (size_t)i * 22;
  float *q = out12 + (size_t)pos[i] * 12;
  q[0] = ld_f32(p), q[1] = ld_f32(p + 4), q[2] = ld_f32(p + 8), q[3] = 1.f;
  q[4] = 0.f, q[5] = 0.f, q[6] = 0.f, q[7] = 0.f;  // :135-137
  q[8] = ld_f32(p + 12);                            // :134
  q[9] = (float)ld_u32(p + 18) * time_unit_scale * 1.e-9f;
  q[10] = 0.f, q[11] = 0.f;
}

// ---- Velodyne (preprocess.cpp:148-212) ------------------------------------------------------------------------------
// One thread per point of the message. What the handler does before its loop (:161-186: is_first / yaw_fp / yaw_last /
// time_last, given_offset_time, yaw_first, yaw_end, layer_first) feeds nothing - the per-ring yaw interpolation FAST-LIO
// has there was removed from MA-LIO's loop - and is not computed. maximum_time: the curvature may be NEGATIVE (drivers that
// stamp relative to the end of the sweep), so the slots hold an order-preserving code of ALL non-NaN floats (0 = none;
// `maximum_time < NaN` is false in the reference too, :206).
__device__ __forceinline__ u32 f32_order_code(float f) {
  const u32 b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // monotonic over the non-NaN floats, never 0 for them
}
struct VelPoint {
  float x, y, z, intensity, curvature;
};
__device__ __forceinline__ VelPoint vel_point(const unsigned char *__restrict__ data, malio_pc2_layout_t lay, float time_unit_scale, int i) {
  const unsigned char *p = data + (size_t)i * (size_t)lay.point_step;
  VelPoint v;
  v.x = ld_f32(p + lay.off_x), v.y = ld_f32(p + lay.off_y), v.z = ld_f32(p + lay.off_z);  // :196-198
  v.intensity = lay.off_intensity >= 0 ? ld_f32(p + lay.off_intensity) : 0.f;               // :199
  const float time = lay.off_time >= 0 ? ld_f32(p + lay.off_time) : 0.f;
  v.curvature = time * time_unit_scale;                                                     // :200
  return v;
}
__global__ void __launch_bounds__(BLK) k_velodyne_select(const unsigned char *__restrict__ data, int n, malio_pc2_layout_t lay,
                                                         int pfn, double blind, float time_unit_scale, u32 *keep, u32 *tmax_code) {
  int i = blockIdx.x * BLK + threadIdx.x;
  u32 code = 0u;
  bool push = false;
  if (i < n && i % pfn == 0) {  // :202
    const VelPoint v = vel_point(data, lay, time_unit_scale, i);
    if ((double)(v.x * v.x + v.y * v.y + v.z * v.z) > blind * blind) {  // :204 (float sum, compared in double)
      push = true;
      if (v.curvature == v.curvature) code = f32_order_code(v.curvature);  // :206-207
    }
  }
  if (i <= n) keep[i] = push ? 1u : 0u;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) code = max(code, (u32)__shfl_xor((int)code, d));
  if ((threadIdx.x & 63) == 0 && code) atomicMax(&tmax_code[blockIdx.x & 63], code);
}
__global__ void __launch_bounds__(BLK) k_velodyne_emit(const unsigned char *__restrict__ data, int n, malio_pc2_layout_t lay,
                                                       float time_unit_scale, const u32 *__restrict__ keep,
                                                       const u32 *__restrict__ pos, float *out12) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n || !keep[i]) return;
  const VelPoint v = vel_point(data, lay, time_unit_scale, i);
  float *q = out12 + (size_t)pos[i] * 12;
  q[0] = v.x, q[1] = v.y, q[2] = v.z, q[3] = 1.f;
  q[4] = 0.f, q[5] = 0.f, q[6] = 0.f, q[7] = 0.f;  // :193-195
  q[8] = v.intensity, q[9] = v.curvature;
  q[10] = 0.f, q[11] = 0.f;
}

// shared tail: scan of keep flags, emit, copy back; `nidx` = number of candidate indices (keep has nidx + 1 entries)
template <class EmitFn>
int finish_decode(Ctx *c, ArenaScope &sc, u32 *keep, int nidx, u32 *tmax_bits, EmitFn emit, malio_point_t *out, int cap,
                  int *out_n, double *maximum_time, bool order_code = false) {
  u32 *pos = nullptr, *tiles = nullptr;
  MALIO_HIP(sc.get(&pos, (size_t)nidx + 1));
  MALIO_HIP(sc.get(&tiles, (size_t)(nidx + 1 + 1023) / 1024 + 2));
  exclusive_scan_u32(c, keep, pos, tiles, nidx + 1);
  u32 total = 0, tb[64];
  MALIO_HIP(hipMemcpyAsync(&total, pos + nidx, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipMemcpyAsync(tb, tmax_bits, sizeof(tb), hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  *out_n = (int)total;
  if (maximum_time) {
    double mt = -9999;  // preprocess.cpp:78,119
    for (int k = 0; k < 64; k++)
      if (tb[k] != 0u) {  // slots hold (float bits + 1): curvatures >= 0, bits order like values; or f32_order_code (Velodyne)
        const u32 bits = order_code ? ((tb[k] & 0x80000000u) ? (tb[k] & 0x7FFFFFFFu) : ~tb[k]) : tb[k] - 1u;
        float f;
        memcpy(&f, &bits, 4);
        if (mt < (double)f) mt = (double)f;
      }
    *maximum_time = mt;
  }
  const int take = std::min((int)total, cap);
  if (take <= 0) return MALIO_OK;
  float *d_out = nullptr;
  MALIO_HIP(sc.get(&d_out, (size_t)total * 12));
  emit(pos, d_out);
  MALIO_HIP(hipMemcpyAsync(out, d_out, sizeof(float) * 12 * (size_t)take, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  MALIO_HIP(hipGetLastError());
  return MALIO_OK;
}

}  // namespace

int decode_livox(Ctx *c, const unsigned char *rec, int n_rec, int n_scans, int pfn, double blind, int eof_point,
                 malio_point_t *out, int cap, int *out_n, double *maximum_time) {
  MALIO_HIP(hipSetDevice(c->device));
  *out_n = 0;
  if (maximum_time) *maximum_time = -9999;
  const int plsize = n_rec + (eof_point ? 1 : 0);  // msg->point_num
  if (plsize <= 1) return MALIO_OK;
  ArenaScope sc(c->arena);
  unsigned char *d_rec = nullptr;
  u32 *valid = nullptr, *vnum = nullptr, *keep = nullptr, *tiles = nullptr, *tmax = nullptr;
  MALIO_HIP(sc.get(&d_rec, (size_t)n_rec * 19 + 16));
  MALIO_HIP(sc.get(&valid, (size_t)plsize + 1));
  MALIO_HIP(sc.get(&vnum, (size_t)plsize + 1));
  MALIO_HIP(sc.get(&keep, (size_t)plsize + 1));
  MALIO_HIP(sc.get(&tiles, (size_t)(plsize + 1 + 1023) / 1024 + 2));
  MALIO_HIP(sc.get(&tmax, 64));
  MALIO_HIP(hipMemcpyAsync(d_rec, rec, (size_t)n_rec * 19, hipMemcpyHostToDevice, c->stream));
  MALIO_HIP(hipMemsetAsync(tmax, 0, sizeof(u32) * 64, c->stream));
  const int nb = (plsize + 1 + BLK - 1) / BLK;
  hipLaunchKernelGGL(k_livox_valid, dim3(nb), dim3(BLK), 0, c->stream, d_rec, n_rec, plsize, n_scans, valid);
  exclusive_scan_u32(c, valid, vnum, tiles, plsize + 1);
  hipLaunchKernelGGL(k_livox_select, dim3(nb), dim3(BLK), 0, c->stream, d_rec, n_rec, plsize, n_scans, pfn, blind, valid, vnum,
                     keep, tmax);
  auto emit = [&](const u32 *pos, float *d_out) {
    hipLaunchKernelGGL(k_livox_emit, dim3(nb), dim3(BLK), 0, c->stream, d_rec, n_rec, plsize, keep, pos, d_out);
  };
  return finish_decode(c, sc, keep, plsize, tmax, emit, out, cap, out_n, maximum_time);
}

int decode_ouster(Ctx *c, const unsigned char *rec, int n, int pfn, double blind, float time_unit_scale, malio_point_t *out,
                  int cap, int *out_n, double *maximum_time) {
  MALIO_HIP(hipSetDevice(c->device));
  *out_n = 0;
  if (maximum_time) *maximum_time = -9999;
  if (n <= 0) return MALIO_OK;
  ArenaScope sc(c->arena);
  unsigned char *d_rec = nullptr;
  u32 *keep = nullptr, *tmax = nullptr;
  MALIO_HIP(sc.get(&d_rec, (size_t)n * 22 + 16));
  MALIO_HIP(sc.get(&keep, (size_t)n + 1));
  MALIO_HIP(sc.get(&tmax, 64));
  MALIO_HIP(hipMemcpyAsync(d_rec, rec, (size_t)n * 22, hipMemcpyHostToDevice, c->stream));
  MALIO_HIP(hipMemsetAsync(tmax, 0, sizeof(u32) * 64, c->stream));
  const int nb = (n + 1 + BLK - 1) / BLK;
  hipLaunchKernelGGL(k_ouster_select, dim3(nb), dim3(BLK), 0, c->stream, d_rec, n, pfn, blind, time_unit_scale, keep, tmax);
  auto emit = [&](const u32 *pos, float *d_out) {
    hipLaunchKernelGGL(k_ouster_emit, dim3(nb), dim3(BLK), 0, c->stream, d_rec, n, time_unit_scale, keep, pos, d_out);
  };
  return finish_decode(c, sc, keep, n, tmax, emit, out, cap, out_n, maximum_time);
}

int decode_velodyne(Ctx *c, const unsigned char *data, int n, const malio_pc2_layout_t &lay, int pfn, double blind,
                    float time_unit_scale, malio_point_t *out, int cap, int *out_n, double *maximum_time) {
  MALIO_HIP(hipSetDevice(c->device));
  *out_n = 0;
  if (n <= 0) return MALIO_OK;  // :157-158: returns before maximum_time is reset - the caller's value stays
  ArenaScope sc(c->arena);
  unsigned char *d_data = nullptr;
  u32 *keep = nullptr, *tmax = nullptr;
  const size_t bytes = (size_t)n * (size_t)lay.point_step;
  MALIO_HIP(sc.get(&d_data, bytes + 16));
  MALIO_HIP(sc.get(&keep, (size_t)n + 1));
  MALIO_HIP(sc.get(&tmax, 64));
  MALIO_HIP(hipMemcpyAsync(d_data, data, bytes, hipMemcpyHostToDevice, c->stream));
  MALIO_HIP(hipMemsetAsync(tmax, 0, sizeof(u32) * 64, c->stream));
  const int nb = (n + 1 + BLK - 1) / BLK;
  hipLaunchKernelGGL(k_velodyne_select, dim3(nb), dim3(BLK), 0, c->stream, d_data, n, lay, pfn, blind, time_unit_scale, keep, tmax);
  auto emit = [&](const u32 *pos, float *d_out) {
    hipLaunchKernelGGL(k_velodyne_emit, dim3(nb), dim3(BLK), 0, c->stream, d_data, n, lay, time_unit_scale, keep, pos, d_out);
  };
  return finish_decode(c, sc, keep, n, tmax, emit, out, cap, out_n, maximum_time, true);
}

}  // namespace malio
