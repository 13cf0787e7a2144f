// Host side of the iterated error-state Kalman update that drives the fused GPU measurement pass:
// the n x n algebra of esekf::update_iterated_dyn_share_modified
// (/root/reference/MA_LIO/include/IKFoM_toolkit/esekfom/esekfom.hpp:495-721) with
// malio_measure() standing where the reference calls h_dyn_share (esekfom.hpp:512) and consuming the
// reduced H^T R^-1 H / H^T R^-1 h instead of M x C rows (esekfom.hpp:621-637).
// State manifold: pos, rot(SO3), offset_R[L](SO3), offset_T[L], vel, bg, ba, grav(S2, |g| = 9.809),
// tangent layout SURVEY.md §2.1 (src/use-ikfom.hpp:14-27, runtime-parametric in L).
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>
#include "../csrc/malio_internal.hpp"
#include "manifold.hpp"

namespace malio {
using namespace mf;
namespace {

// ---- whole-state boxplus / boxminus ------------------------------------------------------------------
void state_boxplus(malio_state_t &x, int L, const double *dx) {
  for (int k = 0; k < 3; k++) x.pos[k] += dx[k];
  double q[4];
  rotvec_quat(dx + 3, 1.0, q);
  qmul(x.rot, q, x.rot);
  for (int l = 0; l < L; l++) {
    rotvec_quat(dx + 6 + 3 * l, 1.0, q);
    qmul(x.offset_R[l], q, x.offset_R[l]);
  }
  for (int l = 0; l < L; l++)
    for (int k = 0; k < 3; k++) x.offset_T[l][k] += dx[6 + 3 * L + 3 * l + k];
  for (int k = 0; k < 3; k++) x.vel[k] += dx[6 + 6 * L + k], x.bg[k] += dx[9 + 6 * L + k], x.ba[k] += dx[12 + 6 * L + k];
  s2_boxplus(x.grav, dx[15 + 6 * L], dx[16 + 6 * L]);
}
void state_boxminus(const malio_state_t &x, const malio_state_t &o, int L, double *res) {
  auto so3 = [](const double *a, const double *b, double *out) {  // log(b^-1 a)
    double bc[4] = {-b[0], -b[1], -b[2], b[3]}, d[4];
    qmul(bc, a, d);
    quat_rotvec(d, out);
  };
  for (int k = 0; k < 3; k++) res[k] = x.pos[k] - o.pos[k];
  so3(x.rot, o.rot, res + 3);
  for (int l = 0; l < L; l++) so3(x.offset_R[l], o.offset_R[l], res + 6 + 3 * l);
  for (int l = 0; l < L; l++)
    for (int k = 0; k < 3; k++) res[6 + 3 * L + 3 * l + k] = x.offset_T[l][k] - o.offset_T[l][k];
  for (int k = 0; k < 3; k++) {
    res[6 + 6 * L + k] = x.vel[k] - o.vel[k];
    res[9 + 6 * L + k] = x.bg[k] - o.bg[k];
    res[12 + 6 * L + k] = x.ba[k] - o.ba[k];
  }
  s2_boxminus(x.grav, o.grav, res + 15 + 6 * L);
}

// ---- dense helpers (row-major, n <= 41) -----------------------------------------------------------------
using Mat = std::vector<double>;
// Forward and back substitution on all right-hand sides at once, one row operation at a time (see invert_cols). W > 0:
// the row width is a compile-time constant and the row being built lives in registers across its j loop (one store per
// row instead of one per (i, j)); same operations on every entry, in the same order.
template <int W>
static void solve_rows(const double *__restrict A, int n, double *__restrict X, int w_runtime = 0) {
  const int w = W > 0 ? W : w_runtime;
  double acc[W > 0 ? W : 64];
  for (int i = 0; i < n; i++) {
    double *xi = X + (size_t)i * w;
    for (int c = 0; c < w; c++) acc[c] = xi[c];
    for (int j = 0; j < i; j++) {
      const double l = A[i * n + j];
      const double *xj = X + (size_t)j * w;
      for (int c = 0; c < w; c++) acc[c] -= l * xj[c];
    }
    for (int c = 0; c < w; c++) xi[c] = acc[c];
  }
  for (int i = n - 1; i >= 0; i--) {
    double *xi = X + (size_t)i * w;
    for (int c = 0; c < w; c++) acc[c] = xi[c];
    for (int j = n - 1; j > i; j--) {  // j descending: the order of a column-oriented sweep (finished rows are
      const double u = A[i * n + j];   // subtracted from all earlier ones as they complete) - what the device loop runs
      const double *xj = X + (size_t)j * w;
      for (int c = 0; c < w; c++) acc[c] -= u * xj[c];
    }
    const double d = A[i * n + i];
    for (int c = 0; c < w; c++) xi[c] = acc[c] / d;
  }
}

// First w columns of A^-1 (n x w, row-major) by LU with partial pivoting - what Eigen's inverse() does for n > 4 - with
// all right-hand sides advanced together. A is destroyed. w = n: the whole inverse.
bool invert_cols(Mat &A, int n, int w, Mat &X) {
  std::vector<int> piv(n);
  for (int i = 0; i < n; i++) piv[i] = i;
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = std::fabs(A[k * n + k]);
    for (int i = k + 1; i < n; i++)
      if (std::fabs(A[i * n + k]) > best) best = std::fabs(A[i * n + k]), p = i;
    if (best == 0) return false;
    if (p != k) {
      for (int j = 0; j < n; j++) std::swap(A[k * n + j], A[p * n + j]);
      std::swap(piv[k], piv[p]);
    }
    double d = A[k * n + k];
    for (int i = k + 1; i < n; i++) {
      double l = (A[i * n + k] /= d);
      if (l != 0)
        for (int j = k + 1; j < n; j++) A[i * n + j] -= l * A[k * n + j];
    }
  }
  // A^-1 = U^-1 L^-1 P with ALL right-hand sides advanced together, one row operation at a time (row-major, so the
  // inner loops run over contiguous columns and vectorise without reassociation). Every entry sees exactly the
  // operations, in exactly the order, of a column-oriented forward/back substitution (csrc/ieskf_dev.hip lds_invert
  // runs the same sequence on the GPU): same bits, ~5x faster than column by column -
  // at 60 us per GPU pass the 35 x 35 algebra between passes is no longer negligible.
  X.assign((size_t)n * w, 0.0);
  for (int i = 0; i < n; i++)
    if (piv[i] < w) X[(size_t)i * w + piv[i]] = 1.0;
  switch (w) {  // the row width as a compile-time constant: a row of X stays in registers across its j loop
    case 12: solve_rows<12>(A.data(), n, X.data()); break;
    case 18: solve_rows<18>(A.data(), n, X.data()); break;
    case 24: solve_rows<24>(A.data(), n, X.data()); break;
    case 30: solve_rows<30>(A.data(), n, X.data()); break;
    case 23: solve_rows<23>(A.data(), n, X.data()); break;
    case 29: solve_rows<29>(A.data(), n, X.data()); break;
    case 35: solve_rows<35>(A.data(), n, X.data()); break;
    case 41: solve_rows<41>(A.data(), n, X.data()); break;
    default: solve_rows<0>(A.data(), n, X.data(), w); break;
  }
  return true;
}
bool invert(Mat &A, int n) {  // in place
  Mat X;
  if (!invert_cols(A, n, n, X)) return false;
  A.swap(X);
  return true;
}
// rows [idx, idx+d) of Dst <- B * rows of Src (first ncols columns); cols of P <- P cols * B^T
void rows_apply(Mat &Dst, const Mat &Src, int n, int idx, int d, const double *B, int ncols) {
  double t[3];
  for (int c = 0; c < ncols; c++) {
    for (int i = 0; i < d; i++) {
      double s = 0;
      for (int k = 0; k < d; k++) s += B[i * d + k] * Src[(idx + k) * n + c];
      t[i] = s;
    }
    for (int i = 0; i < d; i++) Dst[(idx + i) * n + c] = t[i];
  }
}
void cols_applyT(Mat &P, int n, int idx, int d, const double *B) {
  double t[3];
  for (int r = 0; r < n; r++) {
    for (int j = 0; j < d; j++) {
      double s = 0;
      for (int k = 0; k < d; k++) s += P[r * n + idx + k] * B[j * d + k];
      t[j] = s;
    }
    for (int j = 0; j < d; j++) P[r * n + idx + j] = t[j];
  }
}

}  // namespace

}  // namespace malio (reopened below; step_core needs <functional>)
#include <functional>
namespace malio {

// gain(P_projected, K_h, K_x): fills K_h (n) and K_x[:, 0:C] (n x n, rest zero) for the current pass.
using GainFn = std::function<int(StepPre &, std::vector<double> &, std::vector<double> &)>;

// First half of one iteration of esekfom.hpp:509-720: everything that depends on the iterate alone (:526-572, and the
// first of the two inversions of :621) - a caller that has the GPU busy with the measurement pass of this very iterate
// runs it meanwhile (ieskf_update_gated).
void ieskf_step_pre(int L, const malio_state_t *x, const malio_state_t *x_propagated, const double *P_prop, StepPre &pre,
                    bool with_inverse) {
  const int n = 17 + 6 * L;
  const malio_state_t &x_ = *x;
  Mat &P_ = pre.P_;
  P_.assign(P_prop, P_prop + (size_t)n * n);
  std::vector<double> &dx = pre.dx, &dx_new = pre.dx_new;
  dx.assign(n, 0.0);
  const int s2_idx = 15 + 6 * L;
  state_boxminus(x_, *x_propagated, L, dx.data());  // :526
  dx_new = dx;
  for (int b = 0; b <= L; b++) {  // :534-549
    const int idx = b == 0 ? 3 : 6 + 3 * (b - 1);
    double B[9];
    A_matrix_T(&dx[idx], B);
    double tmp[3];
    for (int a = 0; a < 3; a++) tmp[a] = B[a * 3] * dx_new[idx] + B[a * 3 + 1] * dx_new[idx + 1] + B[a * 3 + 2] * dx_new[idx + 2];
    for (int a = 0; a < 3; a++) dx_new[idx + a] = tmp[a];
    rows_apply(P_, P_, n, idx, 3, B, n);
    cols_applyT(P_, n, idx, 3, B);
  }
  {  // :551-572
    double B[4];
    s2_NxMx(x_.grav, x_propagated->grav, dx[s2_idx], dx[s2_idx + 1], B);
    double a0 = B[0] * dx_new[s2_idx] + B[1] * dx_new[s2_idx + 1], a1 = B[2] * dx_new[s2_idx] + B[3] * dx_new[s2_idx + 1];
    dx_new[s2_idx] = a0, dx_new[s2_idx + 1] = a1;
    rows_apply(P_, P_, n, s2_idx, 2, B, n);
    cols_applyT(P_, n, s2_idx, 2, B);
  }
  pre.inv_state = 0;
  if (with_inverse) {
    pre.Pinv = P_;
    pre.inv_state = invert(pre.Pinv, n) ? 1 : -1;
  }
}

template <int N>
static void posterior_rows(const double *__restrict L_, const double *__restrict K_x, const double *__restrict P_, int C,
                           double *__restrict P_out) {
  double acc[N];
  for (int a = 0; a < N; a++) {
    for (int b = 0; b < N; b++) acc[b] = 0.0;
    for (int k = 0; k < C; k++) {
      const double kx = K_x[a * N + k];
      const double *pk = P_ + (size_t)k * N;
      for (int b = 0; b < N; b++) acc[b] += kx * pk[b];
    }
    for (int b = 0; b < N; b++) P_out[a * N + b] = L_[a * N + b] - acc[b];
  }
}

// Second half, AFTER the measurement pass. Pure host code.
//   i          loop index of esekfom.hpp:509 (-1 .. max_iteration-1)
//   x          in: state the pass was evaluated at; out: x boxplus dx
//   t_io       in/out: number of converged iterations so far (esekfom.hpp:658)
//   converge   out: ekfom_data.converge for the NEXT pass (:649-663)
//   done       out: 1 when the posterior covariance was written to P_out and the loop must stop (:665-718)
//   P_out      done: the posterior; otherwise the PROJECTED P_propagated of this iteration - what the reference's member
//              P_ holds from :531-572 until the next valid iteration overwrites it, and therefore what the filter is
//              left with when the loop runs out on invalid passes (`continue` at :514-517 restores nothing)
static int step_post(int L, int maximum_iter, double limit, int i, malio_state_t *x, const malio_state_t *x_propagated,
                     StepPre &pre, const GainFn &gain, int *t_io, int *converge_out, int *done_out, double *P_out) {
  const int n = 17 + 6 * L, C = 6 * (L + 1);
  malio_state_t &x_ = *x;
  Mat &P_ = pre.P_;
  const std::vector<double> &dx_new = pre.dx_new;
  const int s2_idx = 15 + 6 * L;
  std::vector<double> dx_(n), K_h(n);
  Mat K_x((size_t)n * n, 0.0);
  int rc = gain(pre, K_h, K_x);  // :574-640
  if (rc != MALIO_OK) return rc;
  for (int a = 0; a < n; a++) {  // :642
    double s = K_h[a];
    for (int b = 0; b < n; b++) s += (K_x[a * n + b] - (a == b ? 1.0 : 0.0)) * dx_new[b];
    dx_[a] = s;
  }
  state_boxplus(x_, L, dx_.data());  // :646
  bool converge = true;              // :649-657 (limit[i] = 0.001 unless params.limit says otherwise, esekfom.hpp:160-163)
  for (int a = 0; a < n; a++)
    if (std::fabs(dx_[a]) > limit) {
      converge = false;
      break;
    }
  int t = *t_io;
  if (converge) t++;
  if (!t && i == maximum_iter - 2) converge = true;  // :660-663
  *t_io = t;
  *converge_out = converge ? 1 : 0;
  *done_out = 0;
  if (t > 1 || i == maximum_iter - 1) {  // :665-718
    Mat L_(P_);
    for (int b = 0; b <= L; b++) {
      const int idx = b == 0 ? 3 : 6 + 3 * (b - 1);
      double B[9];
      A_matrix_T(&dx_[idx], B);
      rows_apply(L_, P_, n, idx, 3, B, n);
      rows_apply(K_x, K_x, n, idx, 3, B, C);
      cols_applyT(L_, n, idx, 3, B);
      cols_applyT(P_, n, idx, 3, B);
    }
    {
      double B[4];
      s2_NxMx(x_.grav, x_propagated->grav, dx_[s2_idx], dx_[s2_idx + 1], B);
      rows_apply(L_, P_, n, s2_idx, 2, B, n);
      rows_apply(K_x, K_x, n, s2_idx, 2, B, C);
      cols_applyT(L_, n, s2_idx, 2, B);
      cols_applyT(P_, n, s2_idx, 2, B);
    }
    switch (n) {  // :714, P = L - K_x[:, 0:C] P[0:C, :]  (k ascending per entry, row-wise axpy, the row in registers)
      case 23: posterior_rows<23>(L_.data(), K_x.data(), P_.data(), C, P_out); break;
      case 29: posterior_rows<29>(L_.data(), K_x.data(), P_.data(), C, P_out); break;
      case 35: posterior_rows<35>(L_.data(), K_x.data(), P_.data(), C, P_out); break;
      default: posterior_rows<41>(L_.data(), K_x.data(), P_.data(), C, P_out); break;
    }
    *done_out = 1;
  } else {
    memcpy(P_out, P_.data(), sizeof(double) * (size_t)n * n);
  }
  return MALIO_OK;
}

static int step_core(int L, int maximum_iter, double limit, int i, malio_state_t *x, const malio_state_t *x_propagated,
                     const double *P_prop, const GainFn &gain, int *t_io, int *converge_out, int *done_out,
                     double *P_out) {
  StepPre pre;
  ieskf_step_pre(L, x, x_propagated, P_prop, pre, false);
  return step_post(L, maximum_iter, limit, i, x, x_propagated, pre, gain, t_io, converge_out, done_out, P_out);
}

// K_x[:, 0:C] = Pc (n x C) * HtRinvH (C x C): per entry k ascending, as a dot product would; a row at a time, in registers
template <int C>
static void gain_rows(const double *__restrict Pc, const double *__restrict H, int n, double *__restrict K_x) {
  double acc[C];
  for (int a = 0; a < n; a++) {
    for (int b = 0; b < C; b++) acc[b] = 0.0;
    for (int k = 0; k < C; k++) {
      const double p = Pc[a * C + k];
      const double *hk = H + (size_t)k * C;
      for (int b = 0; b < C; b++) acc[b] += p * hk[b];
    }
    double *kx = K_x + (size_t)a * n;
    for (int b = 0; b < C; b++) kx[b] = acc[b];
  }
}

// esekfom.hpp:621-637 on the reduced normal equations: P_inv = (P^-1 + blk(HtRinvH))^-1,
// K_h = P_inv[:, 0:C] HtRinvh, K_x[:, 0:C] = P_inv[:, 0:C] HtRinvH. Only the first C columns of P_inv are used, so only
// they are solved for (every column of an inverse is an independent right-hand side: same values).
static GainFn normal_eq_gain(int L, const double *HtRinvH, const double *HtRinvh) {
  return [=](StepPre &pre, std::vector<double> &K_h, std::vector<double> &K_x) -> int {
    const int n = 17 + 6 * L, C = 6 * (L + 1);
    if (pre.inv_state == 0) {
      pre.Pinv = pre.P_;
      pre.inv_state = invert(pre.Pinv, n) ? 1 : -1;
    }
    if (pre.inv_state < 0) return MALIO_ERR_BAD_ARG;
    Mat &Pt = pre.Pinv, Pc;
    for (int a = 0; a < C; a++)
      for (int b = 0; b < C; b++) Pt[a * n + b] += HtRinvH[a * C + b];
    if (!invert_cols(Pt, n, C, Pc)) return MALIO_ERR_BAD_ARG;
    for (int a = 0; a < n; a++) {
      double s = 0;
      for (int b = 0; b < C; b++) s += Pc[a * C + b] * HtRinvh[b];
      K_h[a] = s;
    }
    switch (C) {  // (row of K_x in registers across its k loop: compile-time width)
      case 12: gain_rows<12>(Pc.data(), HtRinvH, n, K_x.data()); break;
      case 18: gain_rows<18>(Pc.data(), HtRinvH, n, K_x.data()); break;
      case 24: gain_rows<24>(Pc.data(), HtRinvH, n, K_x.data()); break;
      default: gain_rows<30>(Pc.data(), HtRinvH, n, K_x.data()); break;
    }
    return MALIO_OK;
  };
}

int ieskf_step_post(int L, int maximum_iter, double limit, int i, malio_state_t *x, const malio_state_t *x_propagated,
                    StepPre &pre, const double *HtRinvH, const double *HtRinvh, int *t_io, int *converge_out, int *done_out,
                    double *P_out) {
  return step_post(L, maximum_iter, limit > 0 ? limit : 0.001, i, x, x_propagated, pre, normal_eq_gain(L, HtRinvH, HtRinvh), t_io,
                   converge_out, done_out, P_out);
}

int ieskf_step(int L, int maximum_iter, double limit, int i, malio_state_t *x, const malio_state_t *x_propagated,
               const double *P_prop, const double *HtRinvH, const double *HtRinvh, int *t_io, int *converge_out,
               int *done_out, double *P_out) {
  return step_core(L, maximum_iter, limit > 0 ? limit : 0.001, i, x, x_propagated, P_prop, normal_eq_gain(L, HtRinvH, HtRinvh), t_io,
                   converge_out, done_out, P_out);
}

int ieskf_update(Ctx *c, malio_xchg_t xchg, malio_state_t *xio, double *Pio, double R, int *stats, double *solve_time) {
  // h_dyn_share: the fused pass over this GPU's scan, or - with an exchange - over the scan sharded across the node
  PassFn pass = [c, xchg](const malio_state_t *s, int converge, malio_measure_out_t *mo) -> int {
    return xchg ? malio_measure_node((malio_handle_t)c, xchg, s, converge, mo, nullptr) : malio_measure((malio_handle_t)c, s, converge, mo);
  };
  PassFn rows = [c](const malio_state_t *s, int converge, malio_measure_out_t *mo) -> int {
    return malio_measure((malio_handle_t)c, s, converge, mo);
  };
  // the M x M form (esekfom.hpp:574-582) needs every rank's rows: not a sharded path
  return ieskf_update_fn(c->prm, pass, xchg ? nullptr : &rows, c->N, c->pass_hook, c->pass_hook_user, xio, Pio, R, stats, solve_time);
}

int ieskf_update_fn(const malio_params_t &prm, const PassFn &pass, const PassFn *rows_pass, int Nscan, void (*hook)(int, void *),
                    void *hook_user, malio_state_t *xio, double *Pio, double R, int *stats, double *solve_time) {
  const int L = prm.lid_num, n = 17 + 6 * L, C = 6 * (L + 1), maximum_iter = prm.max_iteration;
  const double limit = prm.limit > 0 ? prm.limit : 0.001;
  malio_state_t x_ = *xio;
  const malio_state_t x_propagated = x_;
  const Mat P_prop(Pio, Pio + (size_t)n * n);
  // The covariance of the iterations goes to a local buffer and reaches the caller's P on the success path only: an early
  // return (MALIO_SMALL_M_FALLBACK when a LATER pass accepts fewer points than states, an error) leaves x and P as they
  // came in - the caller redoes the update from them.
  Mat P_work;
  bool P_written = false;
  int converge = 1, t = 0, passes = 0, searches = 0, lastM = 0;
  double solve = 0;
  malio_measure_out_t mo;
  std::vector<double> rows_hx, rows_h, rows_R;
  for (int i = -1; i < maximum_iter; i++) {  // esekfom.hpp:509
    memset(&mo, 0, sizeof(mo));
    searches += converge ? 1 : 0;
    if (hook) hook(passes, hook_user);
    int rc = pass(&x_, converge, &mo);
    passes++;
    if (rc < 0) return rc;
    if (!mo.valid) continue;  // :514-517
    lastM = mo.M;
    if (!rows_pass && n > mo.M) return MALIO_SMALL_M_FALLBACK;
    auto t0 = std::chrono::steady_clock::now();
    GainFn gain;
    if (n > mo.M) {
      // :574-582 small-M fallback: K = P H^T (H P H^T / R + I)^-1 / R with the scalar R, needs the rows.
      // Same state, converge = 0: the accept flags of the pass above stand, so these are its rows.
      const int M = mo.M;
      rows_hx.assign((size_t)Nscan * C, 0.0), rows_h.assign(Nscan, 0.0), rows_R.assign(Nscan, 0.0);
      malio_measure_out_t mr;
      memset(&mr, 0, sizeof(mr));
      mr.h_x = rows_hx.data(), mr.h = rows_h.data(), mr.R = rows_R.data();
      rc = (*rows_pass)(&x_, 0, &mr);
      if (rc < 0) return rc;
      gain = [&, M](StepPre &pre, std::vector<double> &K_h, std::vector<double> &K_x) -> int {
        const Mat &P_ = pre.P_;
        Mat S((size_t)M * M, 0.0), PHt((size_t)n * M, 0.0);
        for (int a = 0; a < n; a++)
          for (int m = 0; m < M; m++) {
            double s = 0;
            for (int b = 0; b < C; b++) s += P_[a * n + b] * rows_hx[(size_t)m * C + b];
            PHt[a * M + m] = s;
          }
        for (int m = 0; m < M; m++)
          for (int k = 0; k < M; k++) {
            double s = 0;
            for (int b = 0; b < C; b++) s += rows_hx[(size_t)m * C + b] * PHt[b * M + k];
            S[m * M + k] = s / R + (m == k ? 1.0 : 0.0);
          }
        if (!invert(S, M)) return MALIO_ERR_BAD_ARG;
        Mat K((size_t)n * M, 0.0);
        for (int a = 0; a < n; a++)
          for (int k = 0; k < M; k++) {
            double s = 0;
            for (int m = 0; m < M; m++) s += PHt[a * M + m] * S[m * M + k];
            K[a * M + k] = s / R;
          }
        for (int a = 0; a < n; a++) {
          double s = 0;
          for (int m = 0; m < M; m++) s += K[a * M + m] * rows_h[m];
          K_h[a] = s;
          for (int b = 0; b < C; b++) {
            double s2 = 0;
            for (int m = 0; m < M; m++) s2 += K[a * M + m] * rows_hx[(size_t)m * C + b];
            K_x[a * n + b] = s2;
          }
        }
        return MALIO_OK;
      };
    } else {
      gain = normal_eq_gain(L, mo.HtRinvH, mo.HtRinvh);
    }
    int done = 0;
    if (!P_written) P_work.resize((size_t)n * n);
    rc = step_core(L, maximum_iter, limit, i, &x_, &x_propagated, P_prop.data(), gain, &t, &converge, &done, P_work.data());
    solve += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc != MALIO_OK) return rc;
    P_written = true;
    if (done) break;
  }
  // When the loop ran out without `done` (its last pass was invalid), P holds what the last VALID iteration left in
  // the reference's member P_: the projected P_propagated (step_core); with no valid pass at all it is untouched.
  *xio = x_;
  if (P_written) memcpy(Pio, P_work.data(), sizeof(double) * (size_t)n * n);
  if (stats) stats[0] = passes, stats[1] = searches, stats[2] = lastM, stats[3] = t;
  if (solve_time) *solve_time += solve;
  return MALIO_OK;
}

}  // namespace malio
