// SO(3) / S2 helpers shared by the host-side filter code (ieskf.cpp: the iterated update; predict.cpp: the IMU
// propagation step): the pieces of IKFoM's mtk the reference's esekf touches, restated on plain arrays.
// Quaternions are (x, y, z, w); 3x3 matrices row-major. Citations are to
// /root/reference/MA_LIO/include/IKFoM_toolkit/mtk/{src/mtkmath.hpp, types/SOn.hpp, types/S2.hpp}.
#pragma once
#include <cmath>
#include <cstring>
// The same functions serve the device-resident update loop (csrc/ieskf_dev.hip): MALIO_HD marks them for both sides when
// the translation unit is HIP; host/*.cpp are plain C++.
#if defined(__HIP__)
#define MALIO_HD __host__ __device__
#else
#define MALIO_HD
#endif

namespace malio {
namespace mf {

constexpr double TOL = 1e-11;               // MTK::tolerance<double>, mtkmath.hpp:122
constexpr double G_LEN = 98090.0 / 10000.0;  // S2<double, 98090, 10000, 1>, use-ikfom.hpp:8

struct Vec3 {
  double v[3];
};
MALIO_HD inline Vec3 cross3(const double *a, const double *b) {
  return {{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}};
}
// Hamilton product of (x,y,z,w) quaternions
MALIO_HD inline void qmul(const double *a, const double *b, double *r) {
  double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  r[0] = x, r[1] = y, r[2] = z, r[3] = w;
}
MALIO_HD inline void quat_R(const double *q, double R[3][3]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0][0] = 1 - 2 * (y * y + z * z), R[0][1] = 2 * (x * y - w * z), R[0][2] = 2 * (x * z + w * y);
  R[1][0] = 2 * (x * y + w * z), R[1][1] = 1 - 2 * (x * x + z * z), R[1][2] = 2 * (y * z - w * x);
  R[2][0] = 2 * (x * z - w * y), R[2][1] = 2 * (y * z + w * x), R[2][2] = 1 - 2 * (x * x + y * y);
}
// Eigen::Quaternion::toRotationMatrix order, row-major 3x3 as 9 doubles (what the pass kernels' matrix form is built from)
MALIO_HD inline void quat_R_eigen(const double q[4], double R[9]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
         tzz = tz * z;
  R[0] = 1 - (tyy + tzz), R[1] = txy - twz, R[2] = txz + twy;
  R[3] = txy + twz, R[4] = 1 - (txx + tzz), R[5] = tyz - twx;
  R[6] = txz - twy, R[7] = tyz + twx, R[8] = 1 - (txx + tyy);
}
// cos / sinc of sqrt(x2) with the Taylor branch of mtkmath.hpp:142-174
MALIO_HD inline void cos_sinc(double x2, double &c, double &s) {
  const double bound = 1.2207031250000000e-04;  // sqrt(sqrt(DBL_EPSILON))
  if (x2 >= bound) {
    double x = std::sqrt(x2);
    c = std::cos(x), s = std::sin(x) / x;
    return;
  }
  const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
  c = 1., s = 1.;
  double term = -0.5 * x2;
  for (int i = 0; i < 3; ++i) {
    c += term;
    term *= inv[2 * i];
    s += term;
    term *= -inv[2 * i + 1] * x2;
  }
}
// quaternion of the rotation vector `v` scaled by `scale` (MTK::exp with half-angle, SOn.hpp:332-336)
MALIO_HD inline void rotvec_quat(const double *v, double scale, double *q) {
  double h = scale / 2, c, s;
  cos_sinc(h * h * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), c, s);
  q[0] = s * h * v[0], q[1] = s * h * v[1], q[2] = s * h * v[2], q[3] = c;
}
// log of a unit quaternion as a rotation vector (SOn.hpp:341-345 -> mtkmath.hpp:268-288)
MALIO_HD inline void quat_rotvec(const double *q, double *v) {
  double nv = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  if (nv < TOL) nv = TOL;
  double s = 2.0 / nv * std::atan(nv / q[3]);
  v[0] = s * q[0], v[1] = s * q[1], v[2] = s * q[2];
}
MALIO_HD inline void hat3(const double *v, double H[3][3]) {
  H[0][0] = 0, H[0][1] = -v[2], H[0][2] = v[1];
  H[1][0] = v[2], H[1][1] = 0, H[1][2] = -v[0];
  H[2][0] = -v[1], H[2][1] = v[0], H[2][2] = 0;
}
// MTK::A_matrix(v)^T (mtkmath.hpp:235-247), row-major 3x3
MALIO_HD inline void A_matrix_T(const double *v, double At[9]) {
  double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], n = std::sqrt(sq);
  double A[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  if (!(n < TOL)) {
    double H[3][3], H2[3][3];
    hat3(v, H);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) H2[i][j] = H[i][0] * H[0][j] + H[i][1] * H[1][j] + H[i][2] * H[2][j];
    double a = (1 - std::cos(n)) / sq, b = (1 - std::sin(n) / n) / sq;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) A[i][j] += a * H[i][j] + b * H2[i][j];
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) At[i * 3 + j] = A[j][i];
}

// ---- S2 (gravity) : S2.hpp, S2_typ == 1 ------------------------------------------------------------
MALIO_HD inline void s2_Bx(const double *g, double B[3][2]) {  // S2.hpp:225-241
  if (g[0] + G_LEN > TOL) {
    double d = G_LEN + g[0];
    B[0][0] = -g[1], B[0][1] = -g[2];
    B[1][0] = G_LEN - g[1] * g[1] / d, B[1][1] = -g[2] * g[1] / d;
    B[2][0] = -g[2] * g[1] / d, B[2][1] = G_LEN - g[2] * g[2] / d;
    for (int i = 0; i < 3; i++) B[i][0] /= G_LEN, B[i][1] /= G_LEN;
  } else {
    for (int i = 0; i < 3; i++) B[i][0] = 0, B[i][1] = 0;
    B[1][1] = -1, B[2][0] = 1;
  }
}
MALIO_HD inline void s2_boxplus(double *g, double d0, double d1) {  // S2.hpp:136-142
  double B[3][2];
  s2_Bx(g, B);
  double Bu[3] = {B[0][0] * d0 + B[0][1] * d1, B[1][0] * d0 + B[1][1] * d1, B[2][0] * d0 + B[2][1] * d1};
  double q[4], R[3][3];
  rotvec_quat(Bu, 1.0, q);
  quat_R(q, R);
  double r[3];
  for (int i = 0; i < 3; i++) r[i] = R[i][0] * g[0] + R[i][1] * g[1] + R[i][2] * g[2];
  g[0] = r[0], g[1] = r[1], g[2] = r[2];
}
MALIO_HD inline void s2_boxminus(const double *g, const double *o, double res[2]) {  // S2.hpp:144-167
  Vec3 c = cross3(g, o);
  double v_sin = std::sqrt(c.v[0] * c.v[0] + c.v[1] * c.v[1] + c.v[2] * c.v[2]);
  double v_cos = g[0] * o[0] + g[1] * o[1] + g[2] * o[2];
  double theta = std::atan2(v_sin, v_cos);
  if (v_sin < TOL) {
    res[0] = std::fabs(theta) > TOL ? 3.1415926 : 0.0;
    res[1] = 0;
    return;
  }
  double B[3][2];
  s2_Bx(o, B);
  Vec3 hv = cross3(o, g);  // hat(other) * vec
  for (int j = 0; j < 2; j++) res[j] = theta / v_sin * (B[0][j] * hv.v[0] + B[1][j] * hv.v[1] + B[2][j] * hv.v[2]);
}
// res_temp_S2 = Nx(x_.grav) * Mx(x_propagated.grav, delta)   (esekfom.hpp:560-564, S2.hpp:269-290)
MALIO_HD inline void s2_NxMx(const double *g_cur, const double *g_prop, double d0, double d1, double out[4]) {
  double Bc[3][2], Hc[3][3], Nx[2][3];
  s2_Bx(g_cur, Bc);
  hat3(g_cur, Hc);
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 3; j++)
      Nx[i][j] = (Bc[0][i] * Hc[0][j] + Bc[1][i] * Hc[1][j] + Bc[2][i] * Hc[2][j]) / G_LEN / G_LEN;
  double Bp[3][2], Hp[3][3], left[3][3];
  s2_Bx(g_prop, Bp);
  hat3(g_prop, Hp);
  if (std::sqrt(d0 * d0 + d1 * d1) < TOL) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) left[i][j] = -Hp[i][j];
  } else {
    // exp(Bu, scalar(1/2)): 1/2 is INTEGER division == 0 in the reference (S2.hpp:287), so the
    // exponential factor is the identity; only -hat(vec) * A_matrix(Bu)^T * Bx remains.
    double Bu[3] = {Bp[0][0] * d0 + Bp[0][1] * d1, Bp[1][0] * d0 + Bp[1][1] * d1, Bp[2][0] * d0 + Bp[2][1] * d1};
    double At[9];
    A_matrix_T(Bu, At);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        left[i][j] = -(Hp[i][0] * At[0 * 3 + j] + Hp[i][1] * At[1 * 3 + j] + Hp[i][2] * At[2 * 3 + j]);
  }
  double Mx[3][2];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 2; j++) Mx[i][j] = left[i][0] * Bp[0][j] + left[i][1] * Bp[1][j] + left[i][2] * Bp[2][j];
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) out[i * 2 + j] = Nx[i][0] * Mx[0][j] + Nx[i][1] * Mx[1][j] + Nx[i][2] * Mx[2][j];
}

// 3x3 symmetric eigenvalues by cyclic Jacobi, ascending in ev
MALIO_HD inline void sym3_eig_jacobi(double a[3][3], double ev[3]) {
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (a[p][q] == 0) continue;
        double theta = (a[q][q] - a[p][p]) / (2 * a[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
        double cs = 1 / std::sqrt(t * t + 1), sn = t * cs;
        for (int k = 0; k < 3; k++) {
          double akp = a[k][p], akq = a[k][q];
          a[k][p] = cs * akp - sn * akq, a[k][q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < 3; k++) {
          double apk = a[p][k], aqk = a[q][k];
          a[p][k] = cs * apk - sn * aqk, a[q][k] = sn * apk + cs * aqk;
        }
      }
  }
  ev[0] = a[0][0], ev[1] = a[1][1], ev[2] = a[2][2];
  double t;
  if (ev[0] > ev[1]) t = ev[0], ev[0] = ev[1], ev[1] = t;
  if (ev[1] > ev[2]) t = ev[1], ev[1] = ev[2], ev[2] = t;
  if (ev[0] > ev[1]) t = ev[0], ev[0] = ev[1], ev[1] = t;
}
// Smallest and largest eigenvalue of a symmetric 3x3 matrix (a00 a11 a22 a01 a02 a12): sigma_3 / sigma_1 of h_x[:, 0:3]
// is sqrt(l_min / l_max) of N^T N (laserMapping.cpp:746-748 takes it from a JacobiSVD of the M x 3 block).
// Closed form (trigonometric solution of the characteristic cubic) where it is well conditioned - the device-resident
// update loop evaluates this on one GPU lane between two kernels, where a Jacobi iteration (3 divisions and 2 square
// roots per rotation, ~20 rotations) costs microseconds; 0.3 us this way. The cubic loses accuracy only when two
// eigenvalues nearly coincide (acos near +-1: error ~ eps / sqrt(1 - r^2)); inside |r| > 1 - 1e-6 the Jacobi iteration
// is used. Absolute error <= ~1e-12 l_max either way (Jacobi: ~1e-16).
MALIO_HD inline void sym3_eig_minmax(double a00, double a11, double a22, double a01, double a02, double a12, double &lmin,
                                     double &lmax) {
  const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
  if (p1 == 0) {  // diagonal
    lmin = a00 < a11 ? (a00 < a22 ? a00 : a22) : (a11 < a22 ? a11 : a22);
    lmax = a00 > a11 ? (a00 > a22 ? a00 : a22) : (a11 > a22 ? a11 : a22);
    return;
  }
  const double q = (a00 + a11 + a22) / 3;
  const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
  const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2 * p1;
  const double pp = std::sqrt(p2 / 6);
  const double c00 = b00 / pp, c11 = b11 / pp, c22 = b22 / pp, c01 = a01 / pp, c02 = a02 / pp, c12 = a12 / pp;
  const double r = (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02)) / 2;
  if (!(std::fabs(r) < 1 - 1e-6)) {
    double a[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}}, ev[3];
    sym3_eig_jacobi(a, ev);
    lmin = ev[0], lmax = ev[2];
    return;
  }
  const double phi = std::acos(r) / 3;
  lmax = q + 2 * pp * std::cos(phi);
  lmin = q + 2 * pp * std::cos(phi + 2.0943951023931954923);  // + 2 pi / 3
}

// Localization weight (laserMapping.cpp:745-756): w = sigma_3 / sigma_1 of h_x[:, 0:3] = sqrt(l_min / l_max) of N^T N,
// mapped onto [cov_min, cov_max] between the two thresholds and clamped outside. Where the bounds
// l_min <= min diagonal, l_max >= max diagonal (Rayleigh) and Gershgorin's discs already decide the clamp, the
// eigenvalues are not needed (a degenerate scene - where they would be the expensive, nearly coinciding ones - is decided
// by the diagonal alone when its weak direction is near an axis).
MALIO_HD inline double localize_weight(double a00, double a11, double a22, double a01, double a02, double a12,
                                       double thresh_min, double thresh_max, double cov_min, double cov_max) {
  const double dmin = a00 < a11 ? (a00 < a22 ? a00 : a22) : (a11 < a22 ? a11 : a22);
  const double dmax = a00 > a11 ? (a00 > a22 ? a00 : a22) : (a11 > a22 ? a11 : a22);
  if (dmax > 0 && (dmin <= 0 || dmin < 0.99 * thresh_min * thresh_min * dmax)) return cov_min;  // w < thresh_min for sure
  const double r0 = std::fabs(a01) + std::fabs(a02), r1 = std::fabs(a01) + std::fabs(a12), r2 = std::fabs(a02) + std::fabs(a12);
  double glo = a00 - r0, ghi = a00 + r0;
  glo = a11 - r1 < glo ? a11 - r1 : glo, ghi = a11 + r1 > ghi ? a11 + r1 : ghi;
  glo = a22 - r2 < glo ? a22 - r2 : glo, ghi = a22 + r2 > ghi ? a22 + r2 : ghi;
  if (glo > 1.01 * thresh_max * thresh_max * ghi) return cov_max;  // w > thresh_max for sure
  double lmin, lmax;
  sym3_eig_minmax(a00, a11, a22, a01, a02, a12, lmin, lmax);
  double weight = std::sqrt(lmin > 0 ? lmin : 0.0) / std::sqrt(lmax);
  if (weight > thresh_max) return cov_max;
  if (weight < thresh_min) return cov_min;
  return (cov_max - cov_min) * (weight - thresh_min) / (thresh_max - thresh_min) + cov_min;
}

// MTK::A_matrix(v) (mtkmath.hpp:235-247)
MALIO_HD inline void A_matrix(const double *v, double A[3][3]) {
  double At[9];
  A_matrix_T(v, At);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = At[j * 3 + i];
}

}  // namespace mf
}  // namespace malio
