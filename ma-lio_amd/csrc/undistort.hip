// a13/a14: per-raw-point undistortion (ImuProcess::UndistortPcl, /root/reference/MA_LIO/src/IMU_Processing.hpp:475-507)
// on the SE(3) cubic B-spline (BsplineSE3::get_pose, src/BsplineSE3.cpp:84-118; quat_ops.h:190-221).
// One thread per raw point: knot interval by binary search on absolute double timestamps (the reference's
// std::map lookups), the three log_se3 of that interval read from a per-scan table (they only depend on the
// interval; the reference recomputes them per point), three exp_se3 + three 3x4 products in double, Eigen's
// matrix->quaternion conversion, then the rigid compensation of :498-503 in Eigen's quaternion-vector order.
// Also emits, per point, how many IMU stamps lie strictly above its time (D_i); a suffix-min scan turns that into
// the reference's single-step `if` counter (:484-494, the uncertainty-interval index written to `intensity`).
// Algorithmic bytes: 16 B read + 16 B written per raw point.
//
// The kernel is bound by f64 issue (half rate on gfx950), so it is written to need few instructions rather than to
// mirror the reference's expression tree (the acceptance bar is <= 1 float ulp on xyz, tests/test_undistort.py):
//   * per knot interval the host precomputes the twist's angle, unit axis k and k k^T - I (= K^2 of the unit skew matrix):
//     exp(b xi) = I + sin(b th) K + (1 - cos(b th)) K^2 then costs one sincos, 18 multiply/adds for R and as many for V,
//     no matrix product K K and no division by th^2 (quat_ops.h:206-217 forms A, B, C and wskew * wskew per point);
//   * the rotations of :498-503 are applied as matrices (the two constant ones prepared on the host) instead of going
//     through Eigen's matrix -> quaternion conversion and four quaternion-vector products;
//   * the knot times, IMU stamps and interval tables of the scan live in LDS (both binary searches were chains of
//     dependent global loads).
#include <algorithm>
#include <cmath>
#include "malio_internal.hpp"
#include "../host/manifold.hpp"

namespace malio {

struct UndArgs {
  int n;
  const float *in12;  // [n][12] pcl::PointXYZINormal: x y z . nx ny nz . intensity curvature[ms] . .
  float4 *out;       // x y z (w: 1 if compensated, 0 if the spline could not bound the point's time)
  int *D;            // [n] IMU stamps above the point's time (counted down from cov_pointer0)
  double lidar_beg_time;
  const double *knot_t;   // [K]
  const double *knot_T;   // [K][12] rows of R|t
  const double *knot_log; // [K-1][6] log_se3(T_k^-1 T_{k+1})
  int K;
  const double *imu_t;  // [n_imu]
  int n_imu, cov_pointer0;
  double eq[4], et[3];  // extrinsic of this LiDAR (q: x,y,z,w)
  double lq[4], lt[3];  // IMU pose at this LiDAR's scan end
  const double *ivl;    // [K-1][UND_IVL] per-interval twist in axis-angle form (see k_undistort)
  double Re[9], RlT[9];  // rotation matrices of eq and of lq^-1 (row-major)
};
constexpr int UND_IVL = 13;   // theta, k[3], K2 = k k^T - I as xx yy zz xy xz yz, translational part v[3]
constexpr int UND_KMAX = 96;  // knots / IMU stamps the LDS tables hold (a 0.1 s scan has ~15 of each)

struct D3u {
  double x, y, z;
};
__device__ __forceinline__ D3u cross_u(D3u a, D3u b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Eigen::QuaternionBase::_transformVector
__device__ __forceinline__ D3u qrot_u(const double q[4], D3u v) {
  D3u qv{q[0], q[1], q[2]};
  D3u uv = cross_u(qv, v);
  uv = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
  D3u c2 = cross_u(qv, uv);
  return {(v.x + q[3] * uv.x) + c2.x, (v.y + q[3] * uv.y) + c2.y, (v.z + q[3] * uv.z) + c2.z};
}

// exp_se3 (quat_ops.h:190-221) as R (row-major 3x3) and t
__device__ __forceinline__ void exp_se3_d(const double v[6], double R[9], double t[3]) {
  const double wx = v[0], wy = v[1], wz = v[2];
  const double theta = sqrt(wx * wx + wy * wy + wz * wz);
  double A, B, C;
  if (theta < 1e-7) {
    A = 1, B = 0.5, C = 1.0 / 6.0;
  } else {
    A = sin(theta) / theta;
    B = (1 - cos(theta)) / (theta * theta);
    C = (1 - A) / (theta * theta);
  }
  const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double K2[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) K2[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
  double V[9];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const double I = (i % 4 == 0) ? 1.0 : 0.0;
    R[i] = I + A * K[i] + B * K2[i];
    V[i] = I + B * K[i] + C * K2[i];
  }
#pragma unroll
  for (int i = 0; i < 3; i++) t[i] = V[i * 3] * v[3] + V[i * 3 + 1] * v[4] + V[i * 3 + 2] * v[5];
}
// (R,t) <- (R,t) * (Rb,tb)
__device__ __forceinline__ void se3_mul_d(double R[9], double t[3], const double Rb[9], const double tb[3]) {
  double Rn[9], tn[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) Rn[i * 3 + j] = R[i * 3] * Rb[j] + R[i * 3 + 1] * Rb[3 + j] + R[i * 3 + 2] * Rb[6 + j];
    tn[i] = R[i * 3] * tb[0] + R[i * 3 + 1] * tb[1] + R[i * 3 + 2] * tb[2] + t[i];
  }
#pragma unroll
  for (int i = 0; i < 9; i++) R[i] = Rn[i];
#pragma unroll
  for (int i = 0; i < 3; i++) t[i] = tn[i];
}

// general-size fallback (more knots or stamps than the LDS tables hold): tables in global memory, the reference's
// expression tree
__global__ void __launch_bounds__(BLK) k_undistort_big(UndArgs a) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= a.n) return;
  const float *pin = a.in12 + (size_t)i * 12;
  const float4 p = make_float4(pin[0], pin[1], pin[2], pin[9]);
  const double point_t = (double)p.w / 1000.0 + a.lidar_beg_time;  // :482
  // D_i: stamps imu_t[k], k <= cov_pointer0, that are > point_t (imu_t ascending)
  {
    int lo = 0, hi = a.cov_pointer0 + 1;  // first index in [0, c0] with imu_t > point_t
    if (hi > a.n_imu) hi = a.n_imu;
    int top = hi;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (a.imu_t[mid] > point_t)
        hi = mid;
      else
        lo = mid + 1;
    }
    a.D[i] = top - lo;
  }
  // knot interval: i1 = (number of knots <= t) - 1; needs i1-1 and i1+2 (BsplineSE3.cpp:173-230)
  int lo = 0, hi = a.K;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a.knot_t[mid] <= point_t)
      lo = mid + 1;
    else
      hi = mid;
  }
  const int i1 = lo - 1;
  if (i1 < 1 || i1 + 2 >= a.K || i == 0) {  // i == 0: the reference's loop stops before begin() (:475-476)
    a.out[i] = make_float4(p.x, p.y, p.z, 0.f);
    return;
  }
  const double t1 = a.knot_t[i1], t2 = a.knot_t[i1 + 1];
  const double DT = t2 - t1;
  const double u = (point_t - t1) / DT;
  const double b0 = 1.0 / 6.0 * (5 + 3 * u - 3 * u * u + u * u * u);
  const double b1 = 1.0 / 6.0 * (1 + 3 * u + 3 * u * u - 2 * u * u * u);
  const double b2 = 1.0 / 6.0 * (u * u * u);
  double R[9], t[3];
  {
    const double *T0 = a.knot_T + (size_t)(i1 - 1) * 12;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      R[r * 3] = T0[r * 4], R[r * 3 + 1] = T0[r * 4 + 1], R[r * 3 + 2] = T0[r * 4 + 2];
      t[r] = T0[r * 4 + 3];
    }
  }
  const double bb[3] = {b0, b1, b2};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double *lg = a.knot_log + (size_t)(i1 - 1 + k) * 6;
    double v[6], Rk[9], tk[3];
#pragma unroll
    for (int c = 0; c < 6; c++) v[c] = bb[k] * lg[c];
    exp_se3_d(v, Rk, tk);
    se3_mul_d(R, t, Rk, tk);  // pose0 * A0 * A1 * A2 (:111)
  }
  // Eigen quaternion-from-matrix (q_GtoI = R_GtoI, :113)
  double q[4];
  {
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
      double s = sqrt(tr + 1.0);
      q[3] = 0.5 * s;
      s = 0.5 / s;
      q[0] = (R[7] - R[5]) * s, q[1] = (R[2] - R[6]) * s, q[2] = (R[3] - R[1]) * s;
    } else {
      int ii = 0;
      if (R[4] > R[0]) ii = 1;
      if (R[8] > (ii == 0 ? R[0] : R[4])) ii = 2;
      const int jj = (ii + 1) % 3, kk = (jj + 1) % 3;
      double s = sqrt(R[ii * 4] - R[jj * 4] - R[kk * 4] + 1.0);
      double qi = 0.5 * s;
      s = 0.5 / s;
      q[3] = (R[kk * 3 + jj] - R[jj * 3 + kk]) * s;
      double qj = (R[jj * 3 + ii] + R[ii * 3 + jj]) * s, qk = (R[kk * 3 + ii] + R[ii * 3 + kk]) * s;
      q[0] = ii == 0 ? qi : (jj == 0 ? qj : qk);
      q[1] = ii == 1 ? qi : (jj == 1 ? qj : qk);
      q[2] = ii == 2 ? qi : (jj == 2 ? qj : qk);
    }
  }
  // :498-503
  const D3u P_i{(double)p.x, (double)p.y, (double)p.z};
  const D3u T_ei{t[0] - a.lt[0], t[1] - a.lt[1], t[2] - a.lt[2]};
  const double eqc[4] = {-a.eq[0], -a.eq[1], -a.eq[2], a.eq[3]};
  const double lqc[4] = {-a.lq[0], -a.lq[1], -a.lq[2], a.lq[3]};
  D3u x = qrot_u(a.eq, P_i);
  x = {x.x + a.et[0], x.y + a.et[1], x.z + a.et[2]};
  x = qrot_u(q, x);
  x = {x.x + T_ei.x, x.y + T_ei.y, x.z + T_ei.z};
  x = qrot_u(lqc, x);
  x = {x.x - a.et[0], x.y - a.et[1], x.z - a.et[2]};
  x = qrot_u(eqc, x);
  a.out[i] = make_float4((float)x.x, (float)x.y, (float)x.z, 1.f);
}

__device__ __forceinline__ void mat3_apply(const double *R, double &x, double &y, double &z) {
  const double a = R[0] * x + R[1] * y + R[2] * z, b = R[3] * x + R[4] * y + R[5] * z, c = R[6] * x + R[7] * y + R[8] * z;
  x = a, y = b, z = c;
}
__device__ __forceinline__ void mat3T_apply(const double *R, double &x, double &y, double &z) {
  const double a = R[0] * x + R[3] * y + R[6] * z, b = R[1] * x + R[4] * y + R[7] * z, c = R[2] * x + R[5] * y + R[8] * z;
  x = a, y = b, z = c;
}
// sin and cos of an angle in [0, pi/4] without the argument reduction of the library's sincos (three calls per point,
// ~80 f64 instructions each, were a quarter of k_undistort): the classical minimax kernels on that interval (odd
// polynomial of degree 13 / even of degree 14, the coefficient set of fdlibm's k_sin.c / k_cos.c; error below one ulp
// as written, without fused multiply-adds). Larger angles - a spline whose knots are more than 45 degrees apart - take the
// library call.
__device__ __forceinline__ void sincos_small(double x, double *sn, double *cs) {
  const double z = x * x;
  {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    *sn = x + v * (S1 + z * r);
  }
  {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    // 1 - (z/2 - z r), with a quarter of x moved out of the subtraction above 0.3 so that it stays exact (k_cos.c)
    const double qx = x < 0.3 ? 0.0 : (x > 0.78125 ? 0.28125 : __hiloint2double(__double2hiint(x) - 0x00200000, 0));
    const double hz = 0.5 * z - qx;
    const double a = 1.0 - qx;
    *cs = a - (hz - z * r);
  }
}
#ifndef UND_BLK
#define UND_BLK 64
#endif
__global__ void __launch_bounds__(UND_BLK) k_undistort(UndArgs a) {
  // the two search tables in LDS (the binary searches are chains of dependent reads); the pose and twist rows of the
  // interval found are read once per point from global memory - neighbouring points share them, they stay in L1
  // (all tables in LDS: every workgroup then copies 13 doubles per knot first - slower from ~30 knots on)
  __shared__ double s_kt[UND_KMAX], s_it[UND_KMAX];
  for (int e = threadIdx.x; e < a.K; e += UND_BLK) s_kt[e] = a.knot_t[e];
  for (int e = threadIdx.x; e < a.n_imu; e += UND_BLK) s_it[e] = a.imu_t[e];
  __syncthreads();
  const int i = blockIdx.x * UND_BLK + threadIdx.x;
  if (i >= a.n) return;
  const float *pin = a.in12 + (size_t)i * 12;
  const float4 p = make_float4(pin[0], pin[1], pin[2], pin[9]);
  const double point_t = (double)p.w / 1000.0 + a.lidar_beg_time;  // :482
  // Two upper bounds (both tables ascending): D_i counts the stamps imu_t[k], k <= cov_pointer0, that are > point_t;
  // i1 = (number of knots <= point_t) - 1, and the spline needs i1 - 1 and i1 + 2 (BsplineSE3.cpp:173-230). The two
  // binary searches advance in lock-step, branch-free, so that their dependent LDS reads overlap (7 halvings: <= 127).
  const int top = min(a.cov_pointer0 + 1, a.n_imu);
  int lo1 = 0, n1 = top, lo = 0, n2 = a.K;
#pragma unroll
  for (int it = 0; it < 7; it++) {
    const int h1 = n1 >> 1, h2 = n2 >> 1;
    const double v1 = s_it[min(lo1 + h1, UND_KMAX - 1)], v2 = s_kt[min(lo + h2, UND_KMAX - 1)];
    const bool g1 = n1 > 0 && v1 <= point_t, g2 = n2 > 0 && v2 <= point_t;
    lo1 = g1 ? lo1 + h1 + 1 : lo1, n1 = g1 ? n1 - h1 - 1 : h1;
    lo = g2 ? lo + h2 + 1 : lo, n2 = g2 ? n2 - h2 - 1 : h2;
  }
  a.D[i] = top - lo1;
  const int i1 = lo - 1;
  if (i1 < 1 || i1 + 2 >= a.K || i == 0) {  // i == 0: the reference's loop stops before begin() (:475-476)
    a.out[i] = make_float4(p.x, p.y, p.z, 0.f);
    return;
  }
  const double t1 = s_kt[i1], t2 = s_kt[i1 + 1];
  const double u = (point_t - t1) / (t2 - t1);
  const double bb[3] = {1.0 / 6.0 * (5 + 3 * u - 3 * u * u + u * u * u), 1.0 / 6.0 * (1 + 3 * u + 3 * u * u - 2 * u * u * u),
                        1.0 / 6.0 * (u * u * u)};
  // :498-503 with the pose never formed: P_compensate = Re^T (Rl^T (T_i (Re P_i + te) - t_l) - te), where
  // T_i = T0 A0 A1 A2 (:111) is applied to the point factor by factor, innermost first. For a unit axis k,
  // exp(th k) x = x + sin(th) (k x x) + (1 - cos th) (k (k.x) - x), and the translation of exp_se3 is the same form on
  // the twist's translational part with ((1 - cos th) / th, (th - sin th) / th).
  double x = (double)p.x, y = (double)p.y, z = (double)p.z;
  mat3_apply(a.Re, x, y, z);
  x += a.et[0], y += a.et[1], z += a.et[2];
#pragma unroll
  for (int k = 2; k >= 0; k--) {
    const double *iv = a.ivl + (size_t)(i1 - 1 + k) * UND_IVL;
    const double th = bb[k] * iv[0];
    double pa, pb, qa, qb;
    if (th < 1e-7) {  // quat_ops.h:201-204
      pa = th, pb = 0.5 * th * th, qa = 0.5 * th, qb = th * th * (1.0 / 6.0);
    } else {
      double sn, cs;
      if (th <= 0.785398163397448279)
        sincos_small(th, &sn, &cs);
      else
        sincos(th, &sn, &cs);
      const double r = 1.0 / th;
      pa = sn, pb = 1 - cs, qa = pb * r, qb = (th - sn) * r;
    }
    const double kx = iv[1], ky = iv[2], kz = iv[3];
    const double ux = bb[k] * iv[10], uy = bb[k] * iv[11], uz = bb[k] * iv[12];
    // rotation of the point
    const double cx = ky * z - kz * y, cy = kz * x - kx * z, cz = kx * y - ky * x;
    const double dt = kx * x + ky * y + kz * z;
    const double dx = kx * dt - x, dy = ky * dt - y, dz = kz * dt - z;
    // translation V u
    const double ex = ky * uz - kz * uy, ey = kz * ux - kx * uz, ez = kx * uy - ky * ux;
    const double du = kx * ux + ky * uy + kz * uz;
    const double fx = kx * du - ux, fy = ky * du - uy, fz = kz * du - uz;
    x = (x + pa * cx + pb * dx) + (ux + qa * ex + qb * fx);
    y = (y + pa * cy + pb * dy) + (uy + qa * ey + qb * fy);
    z = (z + pa * cz + pb * dz) + (uz + qa * ez + qb * fz);
  }
  {
    const double *T0 = a.knot_T + (size_t)(i1 - 1) * 12;
    const double nx = T0[0] * x + T0[1] * y + T0[2] * z + T0[3], ny = T0[4] * x + T0[5] * y + T0[6] * z + T0[7],
                 nz = T0[8] * x + T0[9] * y + T0[10] * z + T0[11];
    x = nx - a.lt[0], y = ny - a.lt[1], z = nz - a.lt[2];
  }
  mat3_apply(a.RlT, x, y, z);
  x -= a.et[0], y -= a.et[1], z -= a.et[2];
  mat3T_apply(a.Re, x, y, z);
  a.out[i] = make_float4((float)x, (float)y, (float)z, 1.f);
}

int spline_interval(const double *times, int n, double ts);
void spline_interval_logs(const double *poses16, int n, double *logs6);

// ---- the reference's pointer walk (:484-494) on the device ----------------------------------------------------
// Going from the last point to the second, cov_pointer steps down by at most ONE per point, so the interval count
// is A_i = min(D_i, A_{i+1} + 1) with A_n = 0. With B_i = A_i + i this is B_i = min(D_i + i, B_{i+1}), B_n = n: a
// suffix minimum, i.e. an inclusive min-scan over the reversed sequence. intensity <- A_i - 1 (:504); point i opens
// an uncertainty entry when A_i > A_{i+1}, and that entry's number is A_i - 1.
// Two small kernels instead of a library scan (hipcub::DeviceScan: reverse + two scan kernels + the consumer):
// k_und_blockmin leaves the minimum of D_i + i of every block of 256 points; k_und_final first folds the minima of the
// blocks to its right (a few hundred values) into its carry, then runs the suffix minimum inside its own 256 points in
// LDS, and goes on to what it always did with A_i and A_{i+1}.
__global__ void __launch_bounds__(BLK) k_und_blockmin(const int *__restrict__ D, int n, int *bm) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  int v = (i >= 1 && i < n) ? D[i] + i : 0x7FFFFFFF;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d));
  __shared__ int sm[BLK / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    int m = sm[0];
#pragma unroll
    for (int k = 1; k < BLK / 64; k++) m = min(m, sm[k]);
    bm[blockIdx.x] = m;
  }
}
__global__ void __launch_bounds__(BLK) k_und_final(const float *__restrict__ in12, const float4 *__restrict__ und,
                                                   const int *__restrict__ D, const int *__restrict__ bm, int nblk, int n,
                                                   float *out12, int *entry, int entry_cap, int *n_entries, float *entry_pts) {
  __shared__ int s_b[BLK];
  __shared__ int s_carry[BLK / 64];
  const int i = blockIdx.x * BLK + threadIdx.x;
  // carry: B of the first point behind this block = min(n, minima of the blocks to the right)
  int cy = n;
  for (int b = (int)blockIdx.x + 1 + (int)threadIdx.x; b < nblk; b += BLK) cy = min(cy, bm[b]);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cy = min(cy, __shfl_xor(cy, d));
  if ((threadIdx.x & 63) == 0) s_carry[threadIdx.x >> 6] = cy;
  s_b[threadIdx.x] = (i >= 1 && i < n) ? D[i] + i : 0x7FFFFFFF;
  __syncthreads();
  int carry = s_carry[0];
#pragma unroll
  for (int k = 1; k < BLK / 64; k++) carry = min(carry, s_carry[k]);
  // suffix minimum inside the block (Hillis-Steele on 256 values)
#pragma unroll
  for (int d = 1; d < BLK; d <<= 1) {
    const int o = threadIdx.x + d < BLK ? s_b[threadIdx.x + d] : 0x7FFFFFFF;
    __syncthreads();
    s_b[threadIdx.x] = min(s_b[threadIdx.x], o);
    __syncthreads();
  }
  if (i >= n) return;
  float v[12];
#pragma unroll
  for (int k = 0; k < 12; k++) v[k] = in12[(size_t)i * 12 + k];
  if (i >= 1) {
    const int A = min(carry, s_b[threadIdx.x]) - i;
    const int An = min(carry, threadIdx.x + 1 < BLK ? s_b[threadIdx.x + 1] : 0x7FFFFFFF) - (i + 1);  // (0 at i + 1 == n)
    if (i == 1) *n_entries = A;
    const float4 u = und[i];
    if (u.w != 0.f) v[0] = u.x, v[1] = u.y, v[2] = u.z, v[8] = (float)(A - 1);
    if (A > An && A - 1 < entry_cap) {  // this point opens uncertainty entry A - 1
      entry[A - 1] = i;
      if (entry_pts) {
#pragma unroll
        for (int k = 0; k < 12; k++) entry_pts[(size_t)(A - 1) * 12 + k] = v[k];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 12; k++) out12[(size_t)i * 12 + k] = v[k];
}

// Undistort n points (device, 12 floats each) into d_out12 (device). Entry points (descending point index, as the
// reference's walk meets them) and their count go to the host arrays when given.
int undistort_core(Ctx *c, const float *d_in12, int n, double lidar_beg_time, const double *knot_times,
                   const double *knot_poses, int n_knots, const double ext_q[4], const double ext_t[3],
                   const double end_q[4], const double end_t[3], const double *imu_stamps, int n_imu, int cov_pointer0,
                   float *d_out12, int *out_entry_point, int *out_n_entries, malio_point_t *out_entry_pts) {
  // per-scan tables: rows of the control poses and the per-interval log twists
  std::vector<double> T12((size_t)n_knots * 12), logs((size_t)(n_knots - 1) * 6);
  for (int k = 0; k < n_knots; k++)
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 4; cc++) T12[(size_t)k * 12 + r * 4 + cc] = knot_poses[(size_t)k * 16 + r * 4 + cc];
  spline_interval_logs(knot_poses, n_knots, logs.data());
  ArenaScope sc(c->arena);
  float4 *d_und = nullptr;
  int *d_D = nullptr, *d_rev = nullptr, *d_entry = nullptr, *d_ne = nullptr;
  double *d_tab = nullptr;
  // the same twists in axis-angle form (k_undistort): angle, unit axis k, K^2 = k k^T - I, translational part
  std::vector<double> ivl((size_t)(n_knots - 1) * UND_IVL, 0.0);
  for (int k = 0; k + 1 < n_knots; k++) {
    const double *lg = &logs[(size_t)k * 6];
    double *iv = &ivl[(size_t)k * UND_IVL];
    const double th = std::sqrt(lg[0] * lg[0] + lg[1] * lg[1] + lg[2] * lg[2]);
    double kx = 0, ky = 0, kz = 0;
    if (th > 1e-300) kx = lg[0] / th, ky = lg[1] / th, kz = lg[2] / th;
    iv[0] = th, iv[1] = kx, iv[2] = ky, iv[3] = kz;
    iv[4] = -(ky * ky + kz * kz), iv[5] = -(kx * kx + kz * kz), iv[6] = -(kx * kx + ky * ky);  // (unit axis: k k^T - I)
    iv[7] = kx * ky, iv[8] = kx * kz, iv[9] = ky * kz;
    iv[10] = lg[3], iv[11] = lg[4], iv[12] = lg[5];
  }
  const size_t ntab = (size_t)n_knots + T12.size() + logs.size() + (size_t)n_imu + ivl.size();
  const int entry_cap = n_imu + 4;
  MALIO_HIP(sc.get(&d_und, (size_t)n));
  MALIO_HIP(sc.get(&d_D, (size_t)n));
  MALIO_HIP(sc.get(&d_rev, (size_t)(n + BLK - 1) / BLK + 1));  // block minima of D_i + i
  MALIO_HIP(sc.get(&d_entry, (size_t)entry_cap));
  MALIO_HIP(sc.get(&d_ne, 1));
  float *d_entry_pts = nullptr;  // the undistorted points that open the entries, for a caller that keeps the cloud in HBM
  if (out_entry_pts) MALIO_HIP(sc.get(&d_entry_pts, (size_t)entry_cap * 12));
  MALIO_HIP(sc.get(&d_tab, ntab));
  std::vector<double> tab;
  tab.insert(tab.end(), knot_times, knot_times + n_knots);
  tab.insert(tab.end(), T12.begin(), T12.end());
  tab.insert(tab.end(), logs.begin(), logs.end());
  tab.insert(tab.end(), imu_stamps, imu_stamps + n_imu);
  tab.insert(tab.end(), ivl.begin(), ivl.end());
  MALIO_HIP(hipMemcpyAsync(d_tab, tab.data(), sizeof(double) * ntab, hipMemcpyHostToDevice, c->stream));
  MALIO_HIP(hipMemsetAsync(d_ne, 0, sizeof(int), c->stream));
  UndArgs a;
  a.n = n, a.in12 = d_in12, a.out = d_und, a.D = d_D, a.lidar_beg_time = lidar_beg_time;
  a.knot_t = d_tab, a.knot_T = d_tab + n_knots, a.knot_log = d_tab + n_knots + T12.size(), a.K = n_knots;
  a.imu_t = d_tab + n_knots + T12.size() + logs.size(), a.n_imu = n_imu, a.cov_pointer0 = cov_pointer0;
  for (int k = 0; k < 4; k++) a.eq[k] = ext_q[k], a.lq[k] = end_q[k];
  for (int k = 0; k < 3; k++) a.et[k] = ext_t[k], a.lt[k] = end_t[k];
  a.ivl = a.imu_t + n_imu;
  mf::quat_R_eigen(ext_q, a.Re);
  {
    double Rl[9];
    mf::quat_R_eigen(end_q, Rl);
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 3; cc++) a.RlT[r * 3 + cc] = Rl[cc * 3 + r];
  }
  const int nb = (n + BLK - 1) / BLK;
  prof_begin(c);
  if (n_knots <= UND_KMAX && n_imu <= UND_KMAX)
    hipLaunchKernelGGL(k_undistort, dim3((n + UND_BLK - 1) / UND_BLK), dim3(UND_BLK), 0, c->stream, a);
  else
    hipLaunchKernelGGL(k_undistort_big, dim3(nb), dim3(BLK), 0, c->stream, a);
  prof_mark(c, "k_undistort");
  hipLaunchKernelGGL(k_und_blockmin, dim3(nb), dim3(BLK), 0, c->stream, (const int *)d_D, n, d_rev);
  hipLaunchKernelGGL(k_und_final, dim3(nb), dim3(BLK), 0, c->stream, d_in12, d_und, (const int *)d_D, (const int *)d_rev, nb, n,
                     d_out12, d_entry, entry_cap, d_ne, d_entry_pts);
  // read-backs through the pinned buffer: [64] count, [65 ..) entry indices, then the entry points
  u32 *mb = nullptr;
  MALIO_HIP(mbox(c, &mb));
  if (65 + (size_t)entry_cap * 13 > MBOX_WORDS) {
    c->err = "malio_undistort: too many IMU stamps for one scan";
    return MALIO_ERR_BAD_ARG;
  }
  int *h_ne = (int *)(mb + 64), *h_ent = (int *)(mb + 65);
  float *h_pts = (float *)(mb + 65 + entry_cap);
  MALIO_HIP(hipMemcpyAsync(h_ne, d_ne, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipMemcpyAsync(h_ent, d_entry, sizeof(int) * entry_cap, hipMemcpyDeviceToHost, c->stream));
  if (out_entry_pts)
    MALIO_HIP(hipMemcpyAsync(h_pts, d_entry_pts, sizeof(float) * 12 * entry_cap, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  prof_end(c);
  MALIO_HIP(hipGetLastError());
  int ne = *h_ne;
  if (ne > entry_cap) ne = entry_cap;
  if (out_entry_point)
    for (int k = 0; k < ne; k++) out_entry_point[k] = h_ent[k];
  if (out_entry_pts) memcpy(out_entry_pts, h_pts, sizeof(malio_point_t) * (size_t)ne);
  if (out_n_entries) *out_n_entries = ne;
  return MALIO_OK;
}

}  // namespace malio

using namespace malio;

extern "C" int malio_undistort(malio_handle_t h, malio_point_t *pts, int n, double lidar_beg_time,
                               const double *knot_times, const double *knot_poses, int n_knots, const double ext_q[4],
                               const double ext_t[3], const double end_q[4], const double end_t[3],
                               const double *imu_stamps, int n_imu, int cov_pointer0, int *out_entry_point,
                               int *out_n_entries) {
  if (!h || !pts || n <= 0 || !knot_times || !knot_poses || n_knots < 4 || !ext_q || !ext_t || !end_q || !end_t ||
      !imu_stamps || n_imu <= 0)
    return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  MALIO_HIP(hipSetDevice(c->device));
  ArenaScope sc(c->arena);
  float *d_in = nullptr, *d_out = nullptr;
  MALIO_HIP(sc.get(&d_in, (size_t)n * 12));
  MALIO_HIP(sc.get(&d_out, (size_t)n * 12));
  MALIO_HIP(hipMemcpyAsync(d_in, pts, sizeof(float) * 12 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  int rc = undistort_core(c, d_in, n, lidar_beg_time, knot_times, knot_poses, n_knots, ext_q, ext_t, end_q, end_t, imu_stamps,
                          n_imu, cov_pointer0, d_out, out_entry_point, out_n_entries, nullptr);
  if (rc != MALIO_OK) return rc;
  // x, y, z and intensity are rewritten in place like :501-504 (the other fields come back unchanged)
  MALIO_HIP(hipMemcpyAsync(pts, d_out, sizeof(float) * 12 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  return MALIO_OK;
}

// Same, but the undistorted cloud of LiDAR `lid` stays in HBM for malio_scan_set_resident (voxel filter + scan upload
// without a host round trip). Only the entry points travel back.
extern "C" int malio_undistort_resident(malio_handle_t h, int lid, const malio_point_t *pts, int n, double lidar_beg_time,
                                        const double *knot_times, const double *knot_poses, int n_knots,
                                        const double ext_q[4], const double ext_t[3], const double end_q[4],
                                        const double end_t[3], const double *imu_stamps, int n_imu, int cov_pointer0,
                                        int *out_entry_point, int *out_n_entries, malio_point_t *out_entry_pts) {
  if (!h || lid < 0 || lid >= h->prm.lid_num || !pts || n <= 0 || !knot_times || !knot_poses || n_knots < 4 || !ext_q ||
      !ext_t || !end_q || !end_t || !imu_stamps || n_imu <= 0)
    return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  MALIO_HIP(hipSetDevice(c->device));
  ResCloud &rcld = c->res[lid];
  if ((size_t)n > rcld.cap) {
    if (rcld.d) (void)hipFree(rcld.d);
    rcld.d = nullptr, rcld.cap = (size_t)n + (size_t)n / 4 + 1024;
    MALIO_HIP(hipMalloc(&rcld.d, sizeof(float) * 12 * rcld.cap));
  }
  rcld.n = 0;
  ArenaScope sc(c->arena);
  float *d_in = nullptr;
  MALIO_HIP(sc.get(&d_in, (size_t)n * 12));
  MALIO_HIP(hipMemcpyAsync(d_in, pts, sizeof(float) * 12 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  int ne = 0;
  std::vector<int> ent((size_t)n_imu + 4);
  int rc = undistort_core(c, d_in, n, lidar_beg_time, knot_times, knot_poses, n_knots, ext_q, ext_t, end_q, end_t, imu_stamps,
                          n_imu, cov_pointer0, rcld.d, ent.data(), &ne, out_entry_pts);  // points: :484-494 needs their pose/time
  if (rc != MALIO_OK) return rc;
  rcld.n = n;
  if (out_entry_point)
    for (int k = 0; k < ne; k++) out_entry_point[k] = ent[k];
  if (out_n_entries) *out_n_entries = ne;
  return MALIO_OK;
}
