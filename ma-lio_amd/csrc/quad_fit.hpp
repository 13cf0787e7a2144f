// esti_plane<float> (common_lib.h:144-190) by the FOUR lanes a query owns in the list walk (round 6).
//
// The plane fit of a search pass - Eigen's ColPivHouseholderQR<Matrix<float, 5, 3>>::solve(-1), the normalisation, the inlier
// test - was ~1 000 dependent instructions of ONE lane per query on the control wave of every workgroup (15 IEEE square roots,
// 26 divisions), a third of the chain every workgroup waits for after its list walks. Here the quad of
// lanes that walked the query's list fits its plane: rows 1..4 of the 5 x 3 system live one per lane (lane s holds row s + 1,
// its right-hand side and, later, its Householder vector entries), row 0 and everything with one value per column (norms,
// pivots, taus, the rows of R) are REPLICATED in the four lanes. A sum over rows is formed left to right as the scalar code
// forms it - row 0's term, then the lanes' terms in order, each fetched by a quad broadcast - so every float operation has the
// operands and the order of oracle/orc_geom.cpp:colpiv_qr_solve_5x3 (the restatement of Eigen 3.3.7 the parity tests pin the
// kernels to): same bits. What the four lanes buy: the four sub-diagonal divisions of a Householder step are ONE division
// (each lane divides its own row's entry), tau rides in the same division on a lane whose row has retired, the trailing
// update and the right-hand side's touch one row per lane, the three column norms are a third each. ~470 instructions per
// lane instead of ~1 000, on all four waves of the workgroup at once (16 queries each).
//
// ONE source for the kernel and for the host: the algorithm is a template over the lane type F (float on the device; a
// struct of four floats on the host, with element-wise operators, where a "broadcast" copies one element to all). The host
// instantiation is what tests/test_quad_fit.py runs against the oracle on millions of neighbour sets, degenerate ones
// included (collinear points, a zero column, ties between column norms, far origins) - the branches a scene rarely takes.
// No data-dependent branch in the algorithm: selects, as the device would predicate anyway.
#pragma once
#include <cmath>

namespace malio {
namespace quad {

#define QF_FN inline

// ---- host backend: four lanes in lockstep (g++ only: the test harness) -------------------------------------------------
#if !defined(__HIPCC__)
struct QB {
  bool v[4];
};
struct QF {
  float v[4];
  QF() = default;
  QF(float x) { v[0] = v[1] = v[2] = v[3] = x; }
};
struct QD {
  double v[4];
  QD() = default;
  QD(double x) { v[0] = v[1] = v[2] = v[3] = x; }
};
#define QF_BIN(T, op)                                          \
  inline T operator op(const T &a, const T &b) {               \
    T r;                                                       \
    for (int i = 0; i < 4; i++) r.v[i] = a.v[i] op b.v[i];     \
    return r;                                                  \
  }
QF_BIN(QF, +) QF_BIN(QF, -) QF_BIN(QF, *) QF_BIN(QF, /) QF_BIN(QD, +) QF_BIN(QD, -) QF_BIN(QD, *) QF_BIN(QD, /)
#undef QF_BIN
#define QF_CMP(T, op)                                          \
  inline QB operator op(const T &a, const T &b) {              \
    QB r;                                                      \
    for (int i = 0; i < 4; i++) r.v[i] = a.v[i] op b.v[i];     \
    return r;                                                  \
  }
QF_CMP(QF, <) QF_CMP(QF, >) QF_CMP(QF, <=) QF_CMP(QF, >=) QF_CMP(QF, ==) QF_CMP(QF, !=) QF_CMP(QD, <) QF_CMP(QD, >) QF_CMP(QD, <=) QF_CMP(QD, >=)
#undef QF_CMP
inline QF operator-(const QF &a) {
  QF r;
  for (int i = 0; i < 4; i++) r.v[i] = -a.v[i];
  return r;
}
inline QB operator&&(const QB &a, const QB &b) {
  QB r;
  for (int i = 0; i < 4; i++) r.v[i] = a.v[i] && b.v[i];
  return r;
}
inline QB operator||(const QB &a, const QB &b) {
  QB r;
  for (int i = 0; i < 4; i++) r.v[i] = a.v[i] || b.v[i];
  return r;
}
inline QB operator!(const QB &a) {
  QB r;
  for (int i = 0; i < 4; i++) r.v[i] = !a.v[i];
  return r;
}
inline QF qsel(const QB &m, const QF &a, const QF &b) {
  QF r;
  for (int i = 0; i < 4; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i];
  return r;
}
inline QD qsel(const QB &m, const QD &a, const QD &b) {
  QD r;
  for (int i = 0; i < 4; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i];
  return r;
}
inline QF qsqrt(const QF &a) {
  QF r;
  for (int i = 0; i < 4; i++) r.v[i] = std::sqrt(a.v[i]);
  return r;
}
inline QF qabs(const QF &a) {
  QF r;
  for (int i = 0; i < 4; i++) r.v[i] = std::fabs(a.v[i]);
  return r;
}
inline QD qabs(const QD &a) {
  QD r;
  for (int i = 0; i < 4; i++) r.v[i] = std::fabs(a.v[i]);
  return r;
}
inline QD qdbl(const QF &a) {
  QD r;
  for (int i = 0; i < 4; i++) r.v[i] = (double)a.v[i];
  return r;
}
inline QF qflt(const QD &a) {
  QF r;
  for (int i = 0; i < 4; i++) r.v[i] = (float)a.v[i];
  return r;
}
template <int S>
inline QF bc(const QF &a) {
  return QF(a.v[S]);
}
template <int S>
inline QD bc(const QD &a) {
  return QD(a.v[S]);
}
template <int S>
inline QB lane_ge(const QF &) {  // lanes S .. 3
  QB r;
  for (int i = 0; i < 4; i++) r.v[i] = i >= S;
  return r;
}
template <int S>
inline QB lane_is(const QF &) {
  QB r;
  for (int i = 0; i < 4; i++) r.v[i] = i == S;
  return r;
}
inline bool qany(const QB &m) { return m.v[0] || m.v[1] || m.v[2] || m.v[3]; }
inline QB qany_quad(const QB &m) {  // true in all four lanes when any lane has it
  const bool a = qany(m);
  QB r;
  for (int i = 0; i < 4; i++) r.v[i] = a;
  return r;
}
#endif

// ---- device backend: one lane of a quad of adjacent lanes --------------------------------------------------------------
// (hipcc parses kernel bodies in its host pass too: there the float "lane type" only has to compile - QF_DEV_ONLY bodies)
#if defined(__HIPCC__)
#undef QF_FN
#define QF_FN __host__ __device__ __forceinline__
template <int S>
QF_FN float bc(float a) {  // quad_perm [S, S, S, S]
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ctrl = S | (S << 2) | (S << 4) | (S << 6);
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), ctrl, 0xF, 0xF, false));
#else
  return a;
#endif
}
template <int S>
QF_FN double bc(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ctrl = S | (S << 2) | (S << 4) | (S << 6);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), ctrl, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), ctrl, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
#else
  return a;
#endif
}
QF_FN float qsel(bool m, float a, float b) { return m ? a : b; }
QF_FN double qsel(bool m, double a, double b) { return m ? a : b; }
QF_FN float qsqrt(float a) { return sqrtf(a); }
QF_FN float qabs(float a) { return fabsf(a); }
QF_FN double qabs(double a) { return fabs(a); }
QF_FN double qdbl(float a) { return (double)a; }
QF_FN float qflt(double a) { return (float)a; }
#if defined(__HIP_DEVICE_COMPILE__)
#define QF_SUB ((int)(threadIdx.x & 3))
#else
#define QF_SUB 0
#endif
template <int S>
QF_FN bool lane_ge(float) {
  return QF_SUB >= S;
}
template <int S>
QF_FN bool lane_is(float) {
  return QF_SUB == S;
}
QF_FN bool qany(bool m) {  // (wave-wide: a superset of "some lane of this quad" - the callers select per lane)
#if defined(__HIP_DEVICE_COMPILE__)
  return __ballot(m) != 0ull;
#else
  return m;
#endif
}
QF_FN bool qany_quad(bool m) {  // any lane of THIS quad
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long b = __ballot(m);
  return ((b >> ((threadIdx.x & 63) & ~3u)) & 0xFull) != 0ull;
#else
  return m;
#endif
}
#endif

// the mask type that comparing two F yields
template <class F>
struct MaskOf {
  typedef decltype(F() < F()) type;
};

// sum over the sub-diagonal rows K + 1 .. 4 (lanes K .. 3) of a per-lane term, left to right from zero - the scalar code's
// `s = 0; for (i = K + 1; i < 5; i++) s += term_i`
template <int K, class F>
QF_FN F sum_rows_from(const F &term) {
  F s = F(0.f);
  if (K <= 0) s = s + bc<0>(term);
  if (K <= 1) s = s + bc<1>(term);
  if (K <= 2) s = s + bc<2>(term);
  s = s + bc<3>(term);
  return s;
}

// Everything the four lanes of a query know about its plane fit when it is over (replicated unless noted)
template <class F>
struct FitState {
  F r[3][3];   // r[k][j]: row k of R (pivot rows, columns in pivot order); r[k][k] = beta_k
  F a[3];      // THIS lane's row (row = lane + 1): Householder vector entries in the columns already eliminated
  F c0, c;     // right-hand side: row 0's (replicated), this lane's row's
  F nU[3], nD[3], hC[3];
  F p[3];      // column permutation as 0.f / 1.f / 2.f: x[p[i]] = y_i
  F nz;        // nonzero_pivots as 0.f .. 3.f
};

// One Householder step of ColPivHouseholderQR::computeInPlace (Eigen/src/QR/ColPivHouseholderQR.h), K = 0, 1, 2
template <int K, class F>
QF_FN void qr_step(FitState<F> &st, const F &threshold_helper, const F &ndt) {
  typedef typename MaskOf<F>::type B;
  F(&r)[3][3] = st.r;
  F(&a)[3] = st.a;
  // ---- pivot: the remaining column of largest updated norm (the first one among equals), swapped into place ----
  constexpr int K1 = K + 1 < 3 ? K + 1 : 2, K2 = K + 2 < 3 ? K + 2 : 2;  // (indices of the branches that exist; the others are never taken)
  F bigv = st.nU[K];
  B m1 = bigv < bigv, m2 = m1;  // (all false) big == K + 1 / big == K + 2
  if (K + 1 < 3) {
    m1 = st.nU[K1] > bigv;
    bigv = qsel(m1, st.nU[K1], bigv);
  }
  if (K + 2 < 3) {
    m2 = st.nU[K2] > bigv;
    bigv = qsel(m2, st.nU[K2], bigv);
    m1 = m1 && !m2;
  }
  const F big_sq = bigv * bigv;
  st.nz = qsel((st.nz == F(3.f)) && (big_sq < threshold_helper * F((float)(5 - K))), F((float)K), st.nz);
  auto swap_if = [](const B &m, F &x, F &y) {
    const F t = x;
    x = qsel(m, y, x);
    y = qsel(m, t, y);
  };
  if (K + 1 < 3) {
    swap_if(m1, a[K], a[K1]);
    for (int i = 0; i <= K && i < 3; i++) swap_if(m1, r[i][K], r[i][K1]);
    swap_if(m1, st.nU[K], st.nU[K1]), swap_if(m1, st.nD[K], st.nD[K1]), swap_if(m1, st.p[K], st.p[K1]);
  }
  if (K + 2 < 3) {
    swap_if(m2, a[K], a[K2]);
    for (int i = 0; i <= K && i < 3; i++) swap_if(m2, r[i][K], r[i][K2]);
    swap_if(m2, st.nU[K], st.nU[K2]), swap_if(m2, st.nD[K], st.nD[K2]), swap_if(m2, st.p[K], st.p[K2]);
  }
  // the pivot row K: row 0 is replicated from the start, row K >= 1 is lane K - 1's, handed to everybody now
  if (K == 1) r[1][1] = bc<0>(a[1]), r[1][2] = bc<0>(a[2]);
  if (K == 2) r[2][2] = bc<1>(a[2]);
  // ---- makeHouseholderInPlace on column K, rows K .. 4 ----
  const B act = lane_ge<K>(a[0]);  // this lane's row is below the pivot row
  const F tailSq = sum_rows_from<K>(a[K] * a[K]);
  const F c0 = r[K][K];
  const B flat = tailSq <= F(1.17549435e-38f);
  F beta = qsqrt(c0 * c0 + tailSq);
  beta = qsel(c0 >= F(0.f), -beta, beta);
  const F den = c0 - beta;
  F tau, v;
  if (K == 0) {  // (all four lanes hold a row below the pivot row: tau has a division of its own)
    v = a[K] / den;
    tau = (beta - c0) / beta;
  } else {  // lane 0's row has retired: it forms tau in the same division
    const B l0 = lane_is<0>(a[0]);
    const F q = qsel(l0, beta - c0, a[K]) / qsel(l0, beta, den);
    tau = bc<0>(q);
    v = q;
  }
  tau = qsel(flat, F(0.f), tau);
  beta = qsel(flat, c0, beta);
  a[K] = qsel(act, qsel(flat, F(0.f), v), a[K]);
  st.hC[K] = tau;
  r[K][K] = beta;
  // ---- applyHouseholderOnTheLeft to the trailing columns ----
  const B upd = tau != F(0.f);
#define QF_APPLY(J)                                                        \
  if ((J) < 3 && (J) > K) {                                                \
    F tmp = sum_rows_from<K>(a[K] * a[(J)]);                               \
    tmp = tmp + r[K][(J)];                                                 \
    r[K][(J)] = qsel(upd, r[K][(J)] - tau * tmp, r[K][(J)]);               \
    a[(J)] = qsel(upd && act, a[(J)] - tau * a[K] * tmp, a[(J)]);          \
  }
  QF_APPLY(1)
  QF_APPLY(2)
#undef QF_APPLY
  // ---- column-norm downdate (LAPACK Working Note 176) ----
#define QF_DOWN(J)                                                                           \
  if ((J) < 3 && (J) > K) {                                                                  \
    const B nzn = st.nU[(J)] != F(0.f);                                                      \
    F temp = qabs(r[K][(J)]) / st.nU[(J)];                                                   \
    temp = (F(1.f) + temp) * (F(1.f) - temp);                                                \
    temp = qsel(temp < F(0.f), F(0.f), temp);                                                \
    const F rr = st.nU[(J)] / st.nD[(J)];                                                    \
    const F temp2 = temp * (rr * rr);                                                        \
    const B redo = nzn && (temp2 <= ndt);                                                    \
    const F scaled = st.nU[(J)] * qsqrt(temp);                                               \
    st.nU[(J)] = qsel(nzn && !redo, scaled, st.nU[(J)]);                                     \
    if (qany(redo)) { /* cancellation: the norm of rows K + 1 .. 4 of the column, afresh */  \
      const F s = qsqrt(sum_rows_from<K>(a[(J)] * a[(J)]));                                  \
      st.nD[(J)] = qsel(redo, s, st.nD[(J)]);                                                \
      st.nU[(J)] = qsel(redo, s, st.nU[(J)]);                                                \
    }                                                                                        \
  }
  QF_DOWN(1)
  QF_DOWN(2)
#undef QF_DOWN
}

// c <- Q^T c for step K (ColPivHouseholderQR::_solve_impl: only the first nonzero_pivots reflections)
template <int K, class F>
QF_FN void rhs_step(FitState<F> &st) {
  typedef typename MaskOf<F>::type B;
  const F tau = st.hC[K];
  const B on = (F((float)K) < st.nz) && (tau != F(0.f));
  F tmp = sum_rows_from<K>(st.a[K] * st.c);
  F ck;
  if (K == 0) ck = st.c0;
  if (K == 1) ck = bc<0>(st.c);
  if (K == 2) ck = bc<1>(st.c);
  tmp = tmp + ck;
  if (K == 0) {
    st.c0 = qsel(on, st.c0 - tau * tmp, st.c0);
  } else {  // the pivot row's right-hand side lives in lane K - 1
    const B holder = K == 1 ? lane_is<0>(tau) : lane_is<1>(tau);
    st.c = qsel(on && holder, st.c - tau * tmp, st.c);
  }
  st.c = qsel(on && lane_ge<K>(tau), st.c - tau * st.a[K] * tmp, st.c);
}

// A(5 x 3) x = -1 in the least-squares sense, as A.colPivHouseholderQr().solve(b) (common_lib.h:174).
//   row0[3]: the first neighbour's coordinates (the same values in the four lanes); mine[3]: neighbour (lane + 1)'s
//   x[3] (out, replicated): the solution
template <class F>
QF_FN void qr_solve_quad(const F row0[3], const F mine[3], F x[3]) {
  typedef typename MaskOf<F>::type B;
  FitState<F> st;
  for (int j = 0; j < 3; j++) st.r[0][j] = row0[j], st.a[j] = mine[j], st.p[j] = F((float)j);
  for (int i = 1; i < 3; i++)
    for (int j = 0; j < 3; j++) st.r[i][j] = F(0.f);
  st.c0 = F(-1.f), st.c = F(-1.f), st.nz = F(3.f);
  const F eps = F(1.1920929e-07f);
  for (int k = 0; k < 3; k++) {
    F s = F(0.f);
    s = s + row0[k] * row0[k];
    const F sq = mine[k] * mine[k];
    s = s + bc<0>(sq);
    s = s + bc<1>(sq);
    s = s + bc<2>(sq);
    s = s + bc<3>(sq);
    st.nD[k] = qsqrt(s);
    st.nU[k] = st.nD[k];
  }
  F maxn = st.nU[0];
  maxn = qsel(st.nU[1] > maxn, st.nU[1], maxn);
  maxn = qsel(st.nU[2] > maxn, st.nU[2], maxn);
  const F th = maxn * eps;
  const F threshold_helper = (th * th) / F(5.0f);
  const F ndt = qsqrt(eps);
  qr_step<0>(st, threshold_helper, ndt);
  qr_step<1>(st, threshold_helper, ndt);
  qr_step<2>(st, threshold_helper, ndt);
  rhs_step<0>(st);
  rhs_step<1>(st);
  rhs_step<2>(st);
  // ---- back substitution on the leading nonzero_pivots x nonzero_pivots block of R (everything replicated) ----
  const F c1 = bc<0>(st.c), c2 = bc<1>(st.c);
  const B n3 = st.nz > F(2.5f), n2 = st.nz > F(1.5f), n1 = st.nz > F(0.5f);
  const F y2 = c2 / st.r[2][2];
  F s1 = c1;
  s1 = qsel(n3, s1 - st.r[1][2] * y2, s1);
  const F y1 = s1 / st.r[1][1];
  F s0 = st.c0;
  s0 = qsel(n2, s0 - st.r[0][1] * y1, s0);
  s0 = qsel(n3, s0 - st.r[0][2] * y2, s0);
  const F y0 = s0 / st.r[0][0];
  const F z = F(0.f);
  const F w0 = qsel(n1, y0, z), w1 = qsel(n2, y1, z), w2 = qsel(n3, y2, z);
  // x[perm[i]] = y_i
  for (int d = 0; d < 3; d++) {
    const F fd = F((float)d);
    x[d] = qsel(st.p[0] == fd, w0, qsel(st.p[1] == fd, w1, w2));
  }
}

// esti_plane<float> without its covariance part: pabcd (replicated) and whether all five points lie within `threshold` of
// the plane. pts: row0 = neighbour 0 (replicated), mine = neighbour lane + 1.
template <class F>
QF_FN typename MaskOf<F>::type esti_plane_quad(const F row0[3], const F mine[3], const F &threshold, F pabcd[4]) {
  typedef typename MaskOf<F>::type B;
  F nv[3];
  qr_solve_quad(row0, mine, nv);
  const F n = qsqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);  // :176
  // :177-180: three float quotients and (float)(1.0 / n) - one division of float operands in double rounded to float, i.e.
  // the correctly rounded float quotient (measure.hip: point_phase) - one per lane, in ONE division
  const B l0 = lane_is<0>(n), l1 = lane_is<1>(n), l2 = lane_is<2>(n);
  const F num = qsel(l0, nv[0], qsel(l1, nv[1], qsel(l2, nv[2], F(1.0f))));
  const F q = num / n;
  pabcd[0] = bc<0>(q), pabcd[1] = bc<1>(q), pabcd[2] = bc<2>(q), pabcd[3] = bc<3>(q);
  // :182-188: each lane its own neighbour, everybody neighbour 0
  const B out0 = qabs(pabcd[0] * row0[0] + pabcd[1] * row0[1] + pabcd[2] * row0[2] + pabcd[3]) > threshold;
  const B outm = qabs(pabcd[0] * mine[0] + pabcd[1] * mine[1] + pabcd[2] * mine[2] + pabcd[3]) > threshold;
  return !qany_quad(out0 || outm);
}

// esti_plane's plane_cov (common_lib.h:159-173) by the four lanes: W0 = neighbour 0's normal_y (replicated), Wm = neighbour
// (lane + 1)'s. Sums left to right from zero over k = 0 .. 4; the five quotients are two divisions (the lanes' own, then
// neighbour 0's in all four). D: the double lane type that goes with F.
template <class F, class D>
QF_FN D unit_cov_quad(const D &cov_threshold, const F &W0, const F &Wm) {
  const D w0 = qdbl(W0), wm = qdbl(Wm);
  const D dm = qabs(cov_threshold - wm);
  D cs = D(0.0);
  cs = cs + qabs(cov_threshold - w0);
  cs = cs + bc<0>(dm);
  cs = cs + bc<1>(dm);
  cs = cs + bc<2>(dm);
  cs = cs + bc<3>(dm);
  const D qm = (cov_threshold - wm) / cs, q0 = (cov_threshold - w0) / cs;
  const D tm = qm * qm * wm, t0 = q0 * q0 * w0;
  D u = D(0.0);
  u = u + t0;
  u = u + bc<0>(tm);
  u = u + bc<1>(tm);
  u = u + bc<2>(tm);
  u = u + bc<3>(tm);
  return qsel(w0 > D(0.00001), u, D(0.0));
}

}  // namespace quad
}  // namespace malio
