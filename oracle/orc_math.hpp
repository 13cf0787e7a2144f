// TEST INFRASTRUCTURE — CPU oracle for the MA-LIO measurement-update hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under
// oracle/. The product (ma-lio_amd/) never includes, links or calls this code.
//
// orc_math.hpp: dependency-free restatement of the Eigen 3 / MTK primitives the reference
// calls on the hot path. Eigen is NOT vendored under /root/reference (CMakeLists.txt:55
// `find_package(Eigen3 REQUIRED)`, version unpinned; Ubuntu 20.04 ships 3.3.7), so the
// published algorithms are restated here and parity versus real Eigen is UNPINNED
// (no golden vectors exist in the reference).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace orc {

struct V3 {
  double x = 0, y = 0, z = 0;
  double &operator[](int i) { return (&x)[i]; }
  double operator[](int i) const { return (&x)[i]; }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

struct M3 {
  double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  static M3 I() {
    M3 r;
    r.m[0][0] = r.m[1][1] = r.m[2][2] = 1;
    return r;
  }
};
inline M3 operator*(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += a.m[i][k] * b.m[k][j];
      r.m[i][j] = s;
    }
  return r;
}
inline V3 operator*(const M3 &a, V3 v) {
  return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
          a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
inline M3 operator+(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j];
  return r;
}
inline M3 operator*(double s, const M3 &a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = s * a.m[i][j];
  return r;
}
inline M3 transpose(const M3 &a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
  return r;
}
inline double trace(const M3 &a) { return a.m[0][0] + a.m[1][1] + a.m[2][2]; }
// SKEW_SYM_MATRX (so3_math.h) == MTK::hat (mtkmath.hpp:176-183) == skew_x (quat_ops.h:81-85)
inline M3 hat(V3 v) {
  M3 r;
  r.m[0][1] = -v.z, r.m[0][2] = v.y;
  r.m[1][0] = v.z, r.m[1][2] = -v.x;
  r.m[2][0] = -v.y, r.m[2][1] = v.x;
  return r;
}

// Hamilton quaternion, Eigen storage/semantics (coeffs = x,y,z,w).
struct Q {
  double x = 0, y = 0, z = 0, w = 1;
};
inline Q conj(Q q) { return {-q.x, -q.y, -q.z, q.w}; }
// Eigen::QuaternionBase::operator* (Hamilton product, no normalisation)
inline Q operator*(Q a, Q b) {
  Q r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
// Eigen::QuaternionBase::_transformVector: v + w*uv + q.vec x uv, uv = 2 (q.vec x v)
inline V3 operator*(Q q, V3 v) {
  V3 qv{q.x, q.y, q.z};
  V3 uv = cross(qv, v);
  uv = uv + uv;
  return v + q.w * uv + cross(qv, uv);
}
// Eigen::QuaternionBase::toRotationMatrix
inline M3 toR(Q q) {
  M3 r;
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  r.m[0][0] = 1 - (tyy + tzz), r.m[0][1] = txy - twz, r.m[0][2] = txz + twy;
  r.m[1][0] = txy + twz, r.m[1][1] = 1 - (txx + tzz), r.m[1][2] = tyz - twx;
  r.m[2][0] = txz - twy, r.m[2][1] = tyz + twx, r.m[2][2] = 1 - (txx + tyy);
  return r;
}
// Eigen quaternion-from-matrix assignment (quaternionbase_assign_impl<..., 3, 3>)
inline Q fromR(const M3 &R) {
  Q q;
  double t = trace(R);
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R.m[2][1] - R.m[1][2]) * t;
    q.y = (R.m[0][2] - R.m[2][0]) * t;
    q.z = (R.m[1][0] - R.m[0][1]) * t;
  } else {
    int i = 0;
    if (R.m[1][1] > R.m[0][0]) i = 1;
    if (R.m[2][2] > R.m[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R.m[i][i] - R.m[j][j] - R.m[k][k] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (R.m[k][j] - R.m[j][k]) * t;
    v[j] = (R.m[j][i] + R.m[i][j]) * t;
    v[k] = (R.m[k][i] + R.m[i][k]) * t;
    q.x = v[0], q.y = v[1], q.z = v[2];
  }
  return q;
}

// ---------------------------------------------------------------------------------------
// Dense dynamic matrix (row-major double) — only what esekfom.hpp:495-721 needs.
struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
  double &operator()(int i, int j) { return a[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
  static Mat I(int n) {
    Mat m(n, n);
    for (int i = 0; i < n; i++) m(i, i) = 1;
    return m;
  }
};
inline Mat operator*(const Mat &A, const Mat &B) {
  assert(A.c == B.r);
  Mat C(A.r, B.c);
  for (int i = 0; i < A.r; i++)
    for (int k = 0; k < A.c; k++) {
      double aik = A(i, k);
      if (aik == 0) continue;
      for (int j = 0; j < B.c; j++) C(i, j) += aik * B(k, j);
    }
  return C;
}
inline Mat operator+(const Mat &A, const Mat &B) {
  Mat C = A;
  for (size_t i = 0; i < C.a.size(); i++) C.a[i] += B.a[i];
  return C;
}
inline Mat operator-(const Mat &A, const Mat &B) {
  Mat C = A;
  for (size_t i = 0; i < C.a.size(); i++) C.a[i] -= B.a[i];
  return C;
}
inline Mat transpose(const Mat &A) {
  Mat T(A.c, A.r);
  for (int i = 0; i < A.r; i++)
    for (int j = 0; j < A.c; j++) T(j, i) = A(i, j);
  return T;
}
// Eigen's MatrixBase::inverse() for sizes > 4 == PartialPivLU().inverse():
// LU with partial (row) pivoting, then solve against the identity.
inline Mat inverse(const Mat &A) {
  int n = A.r;
  assert(A.c == n);
  Mat LU = A;
  std::vector<int> piv(n);
  for (int i = 0; i < n; i++) piv[i] = i;
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = std::fabs(LU(k, k));
    for (int i = k + 1; i < n; i++)
      if (std::fabs(LU(i, k)) > best) best = std::fabs(LU(i, k)), p = i;
    if (p != k) {
      for (int j = 0; j < n; j++) std::swap(LU(k, j), LU(p, j));
      std::swap(piv[k], piv[p]);
    }
    double d = LU(k, k);
    for (int i = k + 1; i < n; i++) {
      LU(i, k) /= d;
      double l = LU(i, k);
      if (l == 0) continue;
      for (int j = k + 1; j < n; j++) LU(i, j) -= l * LU(k, j);
    }
  }
  Mat X(n, n);
  for (int col = 0; col < n; col++) {
    std::vector<double> y(n);
    for (int i = 0; i < n; i++) y[i] = (piv[i] == col) ? 1.0 : 0.0;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < i; j++) y[i] -= LU(i, j) * y[j];
    for (int i = n - 1; i >= 0; i--) {
      for (int j = i + 1; j < n; j++) y[i] -= LU(i, j) * y[j];
      y[i] /= LU(i, i);
    }
    for (int i = 0; i < n; i++) X(i, col) = y[i];
  }
  return X;
}

// ---------------------------------------------------------------------------------------
// MTK primitives (IKFoM_toolkit/mtk/src/mtkmath.hpp)
inline double mtk_tol() { return 1e-11; }  // mtkmath.hpp:122
// mtkmath.hpp:142-174
inline std::pair<double, double> cos_sinc_sqrt(double x2) {
  static const double taylor_0_bound = 2.220446049250313e-16;  // boost epsilon<double>
  static const double taylor_2_bound = std::sqrt(taylor_0_bound);
  static const double taylor_n_bound = std::sqrt(taylor_2_bound);
  if (x2 >= taylor_n_bound) {
    double x = std::sqrt(x2);
    return {std::cos(x), std::sin(x) / x};
  }
  static const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
  double cosi = 1., sinc = 1;
  double term = -1 / 2. * x2;
  for (int i = 0; i < 3; ++i) {
    cosi += term;
    term *= inv[2 * i];
    sinc += term;
    term *= -inv[2 * i + 1] * x2;
  }
  return {cosi, sinc};
}
// MTK::exp (mtkmath.hpp:249-256) wrapped as SO3::exp (SOn.hpp:332-336): scale/2 inside.
inline Q so3_exp(V3 v, double scale = 1) {
  double half = scale / 2;
  double norm2 = dot(v, v);
  auto cs = cos_sinc_sqrt(half * half * norm2);
  double mult = cs.second * half;
  return {mult * v.x, mult * v.y, mult * v.z, cs.first};
}
// S2.hpp:287 calls MTK::exp with scalar(1/2) == 0 (integer division): scale 0.
inline Q mtk_exp_scale(V3 v, double scale) {
  double norm2 = dot(v, v);
  auto cs = cos_sinc_sqrt(scale * scale * norm2);
  double mult = cs.second * scale;
  return {mult * v.x, mult * v.y, mult * v.z, cs.first};
}
// SO3::log (SOn.hpp:341-345) -> MTK::log(res, w, vec, 2, true) (mtkmath.hpp:268-288)
inline V3 so3_log(Q q) {
  V3 vec{q.x, q.y, q.z};
  double nv = norm(vec);
  if (nv < mtk_tol()) nv = mtk_tol();
  double s = 2.0 / nv * std::atan(nv / q.w);
  return s * vec;
}
// MTK::A_matrix (mtkmath.hpp:235-247)
inline M3 A_matrix(V3 v) {
  double sq = v.x * v.x + v.y * v.y + v.z * v.z;
  double n = std::sqrt(sq);
  if (n < mtk_tol()) return M3::I();
  M3 h = hat(v);
  return M3::I() + ((1 - std::cos(n)) / sq) * h + ((1 - std::sin(n) / n) / sq) * (h * h);
}

}  // namespace orc
