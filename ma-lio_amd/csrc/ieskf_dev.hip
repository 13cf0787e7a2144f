// Device-resident iterated update: kf.update_iterated_dyn_share_modified (esekfom.hpp:495-721) as ONE chain of kernels.
// The host enqueues, for every possible pass of the loop (maximum_iter + 1 of them), the pass kernels (measure.hip,
// DEV = true instantiations: they read the state and the control words from the DevLoop block) followed by k_ieskf_step,
// one workgroup that runs the n x n filter algebra of the iteration - what host/ieskf.cpp does between two passes of
// the host-driven loop - and decides what the next pass is: search, reuse, or nothing (loop over: every remaining
// kernel of the chain exits at once). One stream synchronisation per update instead of one per pass, no launch/sync
// round trip and no host arithmetic between passes.
//
// The algebra follows host/ieskf.cpp operation by operation (same projections, the same two LU inversions with partial
// pivoting - elimination and forward substitution advance together, the back substitution runs column-oriented - the
// same accumulation orders in the products); what differs is libm (sin / cos / atan / acos of the device library against
// glibc: <= 1-2 ulp each), so states agree to ~1e-15 relative rather than bit for bit.
// esekfom.hpp line numbers as in host/ieskf.cpp.
#include <chrono>
#include <unistd.h>
#include "malio_internal.hpp"
#include <immintrin.h>
#include "../host/manifold.hpp"

namespace malio {
using namespace mf;
#define STAMP(k)                                    \
  do {                                              \
    if (threadIdx.x == 0) dl->stamps[k] = wall_clock64(); \
  } while (0)

constexpr int ST_BLK = 256;
constexpr int ST_WAVES = ST_BLK / 64;
constexpr int NSUM_ = 97;  // == NSUM in measure.hip (entry layout documented there)

struct StepArgs {
  DevLoop *dl;
  double *P_prop;      // [n*n] the propagated covariance the update started from
  double *P_proj;      // [n*n] projected P_propagated of the last valid iteration (what a loop that runs out leaves in P_)
  const double *sums;  // [L][97] of the pass that just ran (k_final_reduce)
  const double *mm;    // its extrema words: [0..3] extrema
  char *out;           // pinned host memory: DevLoop copy, then P (n*n) at OUT_P_OFF
  int last;            // the last pass the host enqueued
};
constexpr size_t OUT_P_OFF = (sizeof(DevLoop) + 255) & ~(size_t)255;

// ---- wave-level helpers -------------------------------------------------------------------------------------------
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// maximum over the 64 lanes (uniform result) on DPP row operations: six dependent steps of ~20 cycles instead of six
// LDS-crossbar shuffles of ~120 - the pivot search of every elimination step sits on the critical path
__device__ __forceinline__ double wave_max_dpp(double v) {
  v = fmax(v, dpp_f64<0xB1, 0xF>(v));   // quad_perm [1,0,3,2]
  v = fmax(v, dpp_f64<0x4E, 0xF>(v));   // quad_perm [2,3,0,1]
  v = fmax(v, dpp_f64<0x141, 0xF>(v));  // row_half_mirror
  v = fmax(v, dpp_f64<0x140, 0xF>(v));  // row_mirror: every lane of a 16-lane row holds the row's maximum
  v = fmax(v, dpp_f64<0x142, 0xA>(v));  // row_bcast:15 into rows 1 and 3
  v = fmax(v, dpp_f64<0x143, 0xC>(v));  // row_bcast:31 into rows 2 and 3: lane 63 holds the maximum of all
  return readlane_f64(v, 63);
}

// ---- n x n inversion in LDS ------------------------------------------------------------------------------------------
// A [n][ns] is destroyed; X [n][ns] must hold the first w columns of the identity; OUT [n][ns] receives the first w
// columns of A^-1 (row i = row i of the inverse). Same arithmetic per entry as invert() in host/ieskf.cpp:
//   * LU with partial pivoting (first row of the largest |a| wins), the multipliers applied to the right-hand sides in
//     the same sweep (forward substitution: per entry j ascending, exactly the column-by-column order);
//   * back substitution column-oriented (per entry j descending, then the division by the diagonal).
// Rows are never moved: every wave keeps, lane = physical row, the row's logical index in a register and all waves
// take the same pivot decisions from the same column read, so an elimination step needs ONE workgroup barrier.
// Work split: lane = row, the columns of [A | X] interleaved over the waves.
__device__ bool lds_invert(double *A, double *X, double *OUT, int *phys, int n, int w, int ns) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int lrow = lane;
  bool ok = true;
  const int Q = n + w;
  for (int k = 0; k < n; k++) {
    __syncthreads();
    double a = 0.0, v = -1.0;
    if (lane < n) {
      a = A[lane * ns + k];
      if (lrow >= k) v = fabs(a);
    }
    const double m = wave_max_dpp(v);
    const unsigned long long cand = __ballot(v == m);
    int pp;
    if (__popcll(cand) == 1) {
      pp = __ffsll((long long)cand) - 1;
    } else {  // equal magnitudes: the first in (logical) row order, as a sequential scan with `>` keeps it
      int key = ((cand >> lane) & 1ull) ? lrow : (1 << 20);
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) key = min(key, __shfl_xor(key, d));
      pp = __ffsll((long long)__ballot(lane < n && lrow == key)) - 1;
    }
    if (!(m > 0.0)) ok = false;  // singular (or NaN): host/ieskf.cpp invert() returns false
    const double d = readlane_f64(a, pp);
    const int lk = __builtin_amdgcn_readlane(lrow, pp);
    if (lrow == k) lrow = lk;
    if (lane == pp) lrow = k;
    if (lane < n && lrow > k) {
      const double l = a / d;
      if (l != 0) {
        const double *pr = A + pp * ns, *px = X + pp * ns;
        double *mr = A + lane * ns, *mx = X + lane * ns;
        for (int q = wv; q < Q; q += ST_WAVES) {
          if (q < n) {
            if (q > k) mr[q] -= l * pr[q];
          } else {
            mx[q - n] -= l * px[q - n];
          }
        }
      }
    }
  }
  __syncthreads();
  if (wv == 0 && lane < n) phys[lrow] = lane;
  __syncthreads();
  // back substitution: at step i row i is final up to its division; the quotient goes to OUT (so that late readers of
  // X still see the undivided row: one barrier per step), then every earlier row takes its update. lane = column.
  for (int i = n - 1; i >= 0; i--) {
    const int pi = phys[i];
    double xi = 0.0;
    if (lane < w) {
      xi = X[pi * ns + lane] / A[pi * ns + i];
      if (wv == 0) OUT[i * ns + lane] = xi;
      for (int li = wv; li < i; li += ST_WAVES) {
        const int r = phys[li];
        X[r * ns + lane] -= A[r * ns + i] * xi;
      }
    }
    __syncthreads();
  }
  return ok;
}

// rows [idx, idx + d) of Dst <- B * rows of Src (first ncols columns): one thread per column
__device__ __forceinline__ void rows_apply_d(double *Dst, const double *Src, int ns, int idx, int d, const double *B, int ncols) {
  const int c = threadIdx.x;
  if (c >= ncols) return;
  double t[3];
  for (int i = 0; i < d; i++) {
    double s = 0;
    for (int k = 0; k < d; k++) s += B[i * d + k] * Src[(idx + k) * ns + c];
    t[i] = s;
  }
  for (int i = 0; i < d; i++) Dst[(idx + i) * ns + c] = t[i];
}
// columns [idx, idx + d) of P <- P columns * B^T: one thread per row
__device__ __forceinline__ void cols_applyT_d(double *P, int n, int ns, int idx, int d, const double *B) {
  const int r = threadIdx.x;
  if (r >= n) return;
  double t[3];
  for (int j = 0; j < d; j++) {
    double s = 0;
    for (int k = 0; k < d; k++) s += P[r * ns + idx + k] * B[j * d + k];
    t[j] = s;
  }
  for (int j = 0; j < d; j++) P[r * ns + idx + j] = t[j];
}

// x [+] dx and x [-] o on the whole state (host/ieskf.cpp state_boxplus / state_boxminus), block b per thread:
// b = 0 position, 1 rotation, 2 .. 1 + L offset_R, 2 + L .. 1 + 2 L offset_T, then vel, bg, ba, gravity
__device__ void state_boxplus_d(malio_state_t &x, int L, const double *dx, int b) {
  double q[4];
  if (b == 0) {
    for (int k = 0; k < 3; k++) x.pos[k] += dx[k];
  } else if (b == 1) {
    rotvec_quat(dx + 3, 1.0, q);
    qmul(x.rot, q, x.rot);
  } else if (b < 2 + L) {
    const int l = b - 2;
    rotvec_quat(dx + 6 + 3 * l, 1.0, q);
    qmul(x.offset_R[l], q, x.offset_R[l]);
  } else if (b < 2 + 2 * L) {
    const int l = b - 2 - L;
    for (int k = 0; k < 3; k++) x.offset_T[l][k] += dx[6 + 3 * L + 3 * l + k];
  } else if (b == 2 + 2 * L) {
    for (int k = 0; k < 3; k++) x.vel[k] += dx[6 + 6 * L + k];
  } else if (b == 3 + 2 * L) {
    for (int k = 0; k < 3; k++) x.bg[k] += dx[9 + 6 * L + k];
  } else if (b == 4 + 2 * L) {
    for (int k = 0; k < 3; k++) x.ba[k] += dx[12 + 6 * L + k];
  } else if (b == 5 + 2 * L) {
    s2_boxplus(x.grav, dx[15 + 6 * L], dx[16 + 6 * L]);
  }
}
__device__ void state_boxminus_d(const malio_state_t &x, const malio_state_t &o, int L, double *res, int b) {
  auto so3 = [](const double *a, const double *bq, double *out) {  // log(b^-1 a)
    double bc[4] = {-bq[0], -bq[1], -bq[2], bq[3]}, d[4];
    qmul(bc, a, d);
    quat_rotvec(d, out);
  };
  if (b == 0) {
    for (int k = 0; k < 3; k++) res[k] = x.pos[k] - o.pos[k];
  } else if (b == 1) {
    so3(x.rot, o.rot, res + 3);
  } else if (b < 2 + L) {
    const int l = b - 2;
    so3(x.offset_R[l], o.offset_R[l], res + 6 + 3 * l);
  } else if (b < 2 + 2 * L) {
    const int l = b - 2 - L;
    for (int k = 0; k < 3; k++) res[6 + 3 * L + 3 * l + k] = x.offset_T[l][k] - o.offset_T[l][k];
  } else if (b == 2 + 2 * L) {
    for (int k = 0; k < 3; k++) res[6 + 6 * L + k] = x.vel[k] - o.vel[k];
  } else if (b == 3 + 2 * L) {
    for (int k = 0; k < 3; k++) res[9 + 6 * L + k] = x.bg[k] - o.bg[k];
  } else if (b == 4 + 2 * L) {
    for (int k = 0; k < 3; k++) res[12 + 6 * L + k] = x.ba[k] - o.ba[k];
  } else if (b == 5 + 2 * L) {
    s2_boxminus(x.grav, o.grav, res + 15 + 6 * L);
  }
}

// the state in the forms the pass kernels read (quaternions for stage 1, rotation matrices for stage 2); the temporal
// compensation entries of qc / pc are per scan and stay as the host filled them
__device__ void refresh_pass_forms(DevLoop *dl, const malio_state_t &s) {
  const int L = dl->L;
  auto Q = [](const double q[4]) { return Q4{q[0], q[1], q[2], q[3]}; };
  auto V = [](const double t[3]) { return D3{t[0], t[1], t[2]}; };
  QuatConst &qc = dl->qc;
  qc.rot = Q(s.rot), qc.pos = V(s.pos);
  qc.q0 = Q(s.offset_R[0]), qc.t0 = V(s.offset_T[0]);
  PassConst &pc = dl->pc;
  quat_R_eigen(s.rot, pc.Rw);
  quat_R_eigen(s.offset_R[0], pc.R0);
  for (int k = 0; k < 3; k++) pc.pw[k] = s.pos[k], pc.t0[k] = s.offset_T[0][k];
  for (int l = 0; l < MALIO_MAX_LIDAR; l++) {
    const int ll = l < L ? l : 0;
    qc.ql[l] = Q(s.offset_R[ll]), qc.tl[l] = V(s.offset_T[ll]);
    quat_R_eigen(s.offset_R[ll], pc.lid[l].Rl);
    for (int k = 0; k < 3; k++) pc.lid[l].tl[k] = s.offset_T[ll][k];
  }
}

// ---- one iteration of esekfom.hpp:509-720 after its measurement pass ----------------------------------------------------
// LDS (dynamic): four n x n matrices + the C x C normal equations + vectors.
__global__ void __launch_bounds__(ST_BLK) k_ieskf_step(StepArgs g) {
  extern __shared__ double lds[];
  DevLoop *dl = g.dl;
  if (dl->done) return;
  const int L = dl->L, n = 17 + 6 * L, C = 6 * (L + 1), ns = n;  // n is odd: conflict-free column walks
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double *MA = lds;               // inversion work matrix; later K_x
  double *MX = MA + n * ns;       // right-hand sides; later L_ of the posterior
  double *MO = MX + n * ns;       // inverse
  double *MP = MO + n * ns;       // P_ (projected P_propagated)
  double *HtH = MP + n * ns;      // [C][C] H^T R^-1 H with the localization weight
  double *Hth = HtH + C * C;      // [C]
  double *dx = Hth + C;           // [n] x [-] x_propagated
  double *dxn = dx + n;           // [n] dx_new
  double *dxu = dxn + n;          // [n] dx_ (the update)
  double *Kh = dxu + n;           // [n]
  double *Bs = Kh + n;            // [(L + 2)][9] projection blocks
  double *scal = Bs + 9 * (MALIO_MAX_LIDAR + 2);  // [0] M, [1] w^2
  int *phys = (int *)(scal + 4);  // [n]
  __shared__ malio_state_t xs;    // the iterate
  __shared__ int s_flag[4];       // [0] inversion failed
  const int nblk = 6 + 2 * L;     // state blocks (state_boxplus_d)
  const int s2_idx = 15 + 6 * L;

  STAMP(0);
  // ---- validity of the pass (laserMapping.cpp:635-639) ----
  if (tid == 0) {
    double M = 0;
    for (int l = 0; l < L; l++) M += g.sums[l * NSUM_ + 96];
    scal[0] = M;
    s_flag[0] = 0;
    xs = dl->x;
  }
  __syncthreads();
  const int M = (int)(scal[0] + 0.5);
  const int i_loop = dl->i;
  const bool was_search = dl->converge != 0;
  bool done = false;
  int converge_next = dl->converge, t = dl->t, status = MALIO_OK;
  if (M >= 1 && n > M) {  // esekfom.hpp:574-582 needs the rows: the host redoes this update on its own loop
    status = MALIO_SMALL_M_FALLBACK;
    done = true;
  } else if (M >= 1) {
    STAMP(1);
    // ---- :526-572  dx = x [-] x_propagated, projection of dx and of P_propagated ----
    if (tid < nblk) state_boxminus_d(xs, dl->x_prop, L, dx, tid);
    for (int e = tid; e < n * n; e += ST_BLK) MP[e] = g.P_prop[e];
    __syncthreads();
    if (tid < n) dxn[tid] = dx[tid];
    if (tid <= L) A_matrix_T(&dx[tid == 0 ? 3 : 6 + 3 * (tid - 1)], Bs + 9 * tid);
    if (tid == L + 1) s2_NxMx(xs.grav, dl->x_prop.grav, dx[s2_idx], dx[s2_idx + 1], Bs + 9 * (L + 1));
    __syncthreads();
    if (tid <= L) {  // dx_new block = B dx block
      const int idx = tid == 0 ? 3 : 6 + 3 * (tid - 1);
      const double *B = Bs + 9 * tid;
      double tmp[3];
      for (int a = 0; a < 3; a++) tmp[a] = B[a * 3] * dxn[idx] + B[a * 3 + 1] * dxn[idx + 1] + B[a * 3 + 2] * dxn[idx + 2];
      for (int a = 0; a < 3; a++) dxn[idx + a] = tmp[a];
    }
    if (tid == L + 1) {
      const double *B = Bs + 9 * (L + 1);
      const double a0 = B[0] * dxn[s2_idx] + B[1] * dxn[s2_idx + 1], a1 = B[2] * dxn[s2_idx] + B[3] * dxn[s2_idx + 1];
      dxn[s2_idx] = a0, dxn[s2_idx + 1] = a1;
    }
    for (int b = 0; b <= L + 1; b++) {
      const int idx = b == 0 ? 3 : (b <= L ? 6 + 3 * (b - 1) : s2_idx), d = b <= L ? 3 : 2;
      rows_apply_d(MP, MP, ns, idx, d, Bs + 9 * b, n);
      __syncthreads();
      cols_applyT_d(MP, n, ns, idx, d, Bs + 9 * b);
      __syncthreads();
    }
    for (int e = tid; e < n * n; e += ST_BLK) g.P_proj[e] = MP[e];  // what P_ holds if the loop ends without a posterior
    STAMP(2);
    // ---- :621  P_^-1 ----
    for (int e = tid; e < n * n; e += ST_BLK) MA[e] = MP[e], MX[e] = (e / ns == e % ns) ? 1.0 : 0.0;
    bool ok = lds_invert(MA, MX, MO, phys, n, n, ns);
    STAMP(3);
    // ---- the reduced normal equations (measure.hip finish_host): C x C from the per-LiDAR 12 x 12 blocks, w_loc ----
    for (int e = tid; e < C * C + C; e += ST_BLK) HtH[e] = 0.0;  // (Hth follows HtH)
    __syncthreads();
    if (tid == 0) {
      double N6[6] = {0, 0, 0, 0, 0, 0};
      for (int l = 0; l < L; l++)
        for (int k = 0; k < 6; k++) N6[k] += g.sums[l * NSUM_ + 90 + k];
      const double wl = localize_weight(N6[0], N6[1], N6[2], N6[3], N6[4], N6[5], dl->loc_thresh_min, dl->loc_thresh_max,
                                        dl->loc_cov_min, dl->loc_cov_max);
      scal[1] = wl * wl;
    }
    // entry (ga, gb) of the C x C matrix: the contributions of the LiDARs in ascending order, as finish_host adds them
    for (int e = tid; e < C * C + C; e += ST_BLK) {
      const bool rhs = e >= C * C;
      const int ga = rhs ? e - C * C : e / C, gb = rhs ? 0 : e % C;
      double s = 0.0;
      for (int l = 0; l < L; l++) {
        auto loc = [&](int gidx) -> int {  // global column -> index inside LiDAR l's 12-block, or -1
          if (gidx < 6) return gidx;
          if (gidx >= 6 + 3 * l && gidx < 9 + 3 * l) return 6 + gidx - (6 + 3 * l);
          if (gidx >= 6 + 3 * (L + l) && gidx < 9 + 3 * (L + l)) return 9 + gidx - (6 + 3 * (L + l));
          return -1;
        };
        const int a = loc(ga);
        if (a < 0) continue;
        const double *sl = g.sums + l * NSUM_;
        if (rhs) {
          s += sl[78 + a];
        } else {
          const int b = loc(gb);
          if (b < 0) continue;
          const int lo = a < b ? a : b, hi = a < b ? b : a;
          s += sl[lo * 12 - lo * (lo - 1) / 2 + (hi - lo)];  // upper triangle, row-major
        }
      }
      HtH[e] = s;
    }
    __syncthreads();
    const double w2 = scal[1];
    for (int e = tid; e < C * C + C; e += ST_BLK) HtH[e] *= w2;
    STAMP(4);
    // ---- :629-633  P_inv = (P_^-1 + blk(HtRinvH))^-1, its first C columns ----
    __syncthreads();
    for (int e = tid; e < n * n; e += ST_BLK) {
      const int a = e / ns, b = e % ns;
      double v = MO[e];
      if (a < C && b < C) v += HtH[a * C + b];
      MA[e] = v;
      MX[e] = (a == b) ? 1.0 : 0.0;
    }
    ok = lds_invert(MA, MX, MO, phys, n, C, ns) && ok;
    if (!ok && tid == 0) s_flag[0] = 1;
    STAMP(5);
    // ---- :635-640  K_h = P_inv[:, 0:C] HtRinvh, K_x[:, 0:C] = P_inv[:, 0:C] HtRinvH  (K_x lives in MA) ----
    if (tid < n) {
      double s = 0;
      for (int b = 0; b < C; b++) s += MO[tid * ns + b] * Hth[b];
      Kh[tid] = s;
    }
    for (int e = tid; e < n * n; e += ST_BLK) {
      const int a = e / ns, b = e % ns;
      double s = 0.0;
      if (b < C)
        for (int k = 0; k < C; k++) s += MO[a * ns + k] * HtH[k * C + b];
      MA[e] = s;
    }
    __syncthreads();
    STAMP(6);
    // ---- :642-663 ----
    if (tid < n) {
      double s = Kh[tid];
      for (int b = 0; b < n; b++) s += (MA[tid * ns + b] - (tid == b ? 1.0 : 0.0)) * dxn[b];
      dxu[tid] = s;
    }
    __syncthreads();
    if (tid < nblk) state_boxplus_d(xs, L, dxu, tid);
    bool converge = true;
    const double limit = dl->limit;
    for (int a = 0; a < n; a++)
      if (fabs(dxu[a]) > limit) converge = false;
    if (converge) t++;
    if (!t && i_loop == dl->maximum_iter - 2) converge = true;
    converge_next = converge ? 1 : 0;
    __syncthreads();
    STAMP(7);
    if (s_flag[0]) {
      status = MALIO_ERR_BAD_ARG;  // singular covariance: what ieskf_update returns
      done = true;
    } else if (t > 1 || i_loop == dl->maximum_iter - 1) {
      // ---- :665-718  posterior: L_ = P_, both projected by the update's blocks, P = L_ - K_x[:, 0:C] P_[0:C, :] ----
      if (tid <= L) A_matrix_T(&dxu[tid == 0 ? 3 : 6 + 3 * (tid - 1)], Bs + 9 * tid);
      if (tid == L + 1) s2_NxMx(xs.grav, dl->x_prop.grav, dxu[s2_idx], dxu[s2_idx + 1], Bs + 9 * (L + 1));
      for (int e = tid; e < n * n; e += ST_BLK) MX[e] = MP[e];
      __syncthreads();
      for (int b = 0; b <= L + 1; b++) {
        const int idx = b == 0 ? 3 : (b <= L ? 6 + 3 * (b - 1) : s2_idx), d = b <= L ? 3 : 2;
        rows_apply_d(MX, MP, ns, idx, d, Bs + 9 * b, n);
        rows_apply_d(MA, MA, ns, idx, d, Bs + 9 * b, C);
        __syncthreads();
        cols_applyT_d(MX, n, ns, idx, d, Bs + 9 * b);
        cols_applyT_d(MP, n, ns, idx, d, Bs + 9 * b);
        __syncthreads();
      }
      double *Pout = (double *)(g.out + OUT_P_OFF);
      if (lane < n)
        for (int a = wv; a < n; a += ST_WAVES) {
          double acc = 0.0;
          for (int k = 0; k < C; k++) acc += MA[a * ns + k] * MP[k * ns + lane];
          Pout[a * n + lane] = MX[a * ns + lane] - acc;
        }
      done = true;
    }
  }
  STAMP(8);
  // ---- control words for the next pass, outputs when the loop ends here ----
  const bool ends = done || g.last || i_loop >= dl->maximum_iter - 1;
  const bool valid = M >= 1 && status == MALIO_OK;
  if (ends && !(done && status == MALIO_OK)) {  // no posterior: P_ of the last valid iteration, or P untouched
    const double *src = (valid || dl->valid_any) ? g.P_proj : g.P_prop;
    double *Pout = (double *)(g.out + OUT_P_OFF);
    __syncthreads();
    __threadfence();
    for (int e = tid; e < n * n; e += ST_BLK) Pout[e] = src[e];
  }
  __syncthreads();
  if (tid == 0) {
    dl->stamps[9] = wall_clock64();
    dl->passes += 1;
    dl->searches += was_search ? 1 : 0;
    if (was_search) dl->search_skip = dl->skip_opt;  // its certificates exist: later search passes may keep neighbours
    dl->last_search = was_search ? 1 : 0;
    dl->commit_prev = M > 0 ? 1 : 0;
    dl->lastM = M;
    if (valid) {
      dl->lastM_valid = M;
      dl->x = xs;
      dl->t = t;
      dl->converge = converge_next;
      dl->valid_any = 1;
      refresh_pass_forms(dl, xs);
    }
    dl->status = status;
    dl->i = i_loop + 1;
    if (ends) {
      dl->done = 1;
    } else {
      dl->mm_parity ^= 1;
    }
    dl->stamps[10] = wall_clock64();
    if (ends) {
      __threadfence();
      DevLoop *o = (DevLoop *)g.out;
      *o = *dl;
    }
  }
}

// first kernel of the chain: the control block and P_propagated come from the host's pinned, device-mapped staging
// buffer (12 KB over PCIe in one small kernel: an SDMA copy of this size has a longer start-up)
__global__ void __launch_bounds__(256) k_loop_init(const double *__restrict__ src, double *dst, int ndoubles) {
  for (int e = threadIdx.x; e < ndoubles; e += 256) dst[e] = src[e];
}

static size_t loop_block_doubles() { return (sizeof(DevLoop) + 255) / 256 * 32; }  // DevLoop rounded up to 256 B

void free_dev_loop(Ctx *c) {
  if (c->d_loopbuf) (void)hipFree(c->d_loopbuf);
  if (c->h_loop_in) (void)hipHostFree(c->h_loop_in);
  if (c->h_loop_out) (void)hipHostFree(c->h_loop_out);
  if (c->h_gate) (void)hipHostFree(c->h_gate);
  if (c->d_gate_ticket) (void)hipFree(c->d_gate_ticket);
  if (c->d_cmd) (void)hipFree(c->d_cmd);
  c->d_gate_ticket = nullptr, c->d_cmd = nullptr;
  c->h_gate = c->d_gate = nullptr;
  c->d_loopbuf = nullptr, c->d_loop = nullptr, c->h_loop_in = nullptr, c->h_loop_out = nullptr, c->d_loop_out = nullptr;
}

// The device-resident loop in two halves: `begin` enqueues the whole update (max_iteration + 1 passes and the n x n algebra
// of every iteration) and returns; `end` waits for it and hands out the results. Between the two the calling thread is
// free - 0.75 ms at BASELINE config 2 - which is what this mode is for: a host that has something better to do than to
// attend a 0.16 ms gated update (malio_update_iterated_begin / _end).
int ieskf_update_device_begin(Ctx *c, const malio_state_t *xio, const double *Pio) {
  const int L = c->prm.lid_num, n = 17 + 6 * L, C = 6 * (L + 1), maximum_iter = c->prm.max_iteration;
  if (int rc = prepare_scan_dev(c, xio)) return rc;
  if (int rc = resolve_scan_segments(c)) return rc;  // the chain's stage-2 kernels are launched per LiDAR segment
  const size_t hdr = loop_block_doubles(), nn = (size_t)DEV_NMAX * DEV_NMAX;
  if (!c->d_loopbuf) {
    MALIO_HIP(hipMalloc(&c->d_loopbuf, sizeof(double) * (hdr + 2 * nn)));
    c->d_loop = reinterpret_cast<DevLoop *>(c->d_loopbuf);
    MALIO_HIP(hipHostMalloc((void **)&c->h_loop_in, sizeof(double) * (hdr + nn), hipHostMallocMapped));
    MALIO_HIP(hipHostMalloc((void **)&c->h_loop_out, OUT_P_OFF + sizeof(double) * nn, hipHostMallocMapped | hipHostMallocCoherent));
    MALIO_HIP(hipHostGetDevicePointer((void **)&c->d_loop_out, c->h_loop_out, 0));
  }
  // ---- the block the loop starts from ----
  DevLoop *in = reinterpret_cast<DevLoop *>(c->h_loop_in);
  memset(in, 0, sizeof(DevLoop));
  in->done = 0, in->converge = 1, in->i = -1, in->t = 0, in->status = MALIO_OK;
  in->mm_parity = c->mm_parity ^ 1;
  in->commit_prev = c->last_M > 0 ? 1 : 0;
  in->maximum_iter = maximum_iter, in->L = L, in->extrinsic_est_en = c->prm.extrinsic_est_en;
  // the loop's first pass is a search pass: it may keep neighbours of an earlier search of this scan (malio_measure before
  // the update); every later search pass of the loop may keep those of the first (k_ieskf_step arms search_skip)
  in->search_skip = search_skip_begin(c), in->skip_opt = c->opt_search_skip | (c->opt_probe_cache ? 6 : 0);  // (bits: search_skip_begin)
  in->limit = c->prm.limit > 0 ? c->prm.limit : 0.001;
  memcpy(in->tcq, c->tcq, sizeof(in->tcq)), memcpy(in->tct, c->tct, sizeof(in->tct));
  in->x = *xio, in->x_prop = *xio;
  fill_quat_const(c, xio, in->qc);
  fill_pass_const(c, xio, in->pc);
  in->loc_thresh_min = c->prm.localize_thresh_min, in->loc_thresh_max = c->prm.localize_thresh_max;
  in->loc_cov_min = c->prm.localize_cov_min, in->loc_cov_max = c->prm.localize_cov_max;
  double *Pin = reinterpret_cast<double *>(c->h_loop_in) + hdr;
  memcpy(Pin, Pio, sizeof(double) * (size_t)n * n);
  double *d_in = nullptr;
  MALIO_HIP(hipHostGetDevicePointer((void **)&d_in, c->h_loop_in, 0));
  const int ndbl = (int)(hdr + (size_t)n * n);
  hipLaunchKernelGGL(k_loop_init, dim3(1), dim3(256), 0, c->stream, (const double *)d_in, c->d_loopbuf, ndbl);
  // ---- the whole loop, enqueued ----
  StepArgs g;
  g.dl = c->d_loop, g.P_prop = c->d_loopbuf + hdr, g.P_proj = c->d_loopbuf + hdr + nn;
  const int ns_ = sums_len(c);
  g.sums = c->d_sums, g.mm = c->d_sums + ns_, g.out = c->d_loop_out;
  const size_t lds_bytes = sizeof(double) * ((size_t)4 * n * n + (size_t)C * C + C + 4 * n + 9 * (MALIO_MAX_LIDAR + 2) + 4) + sizeof(int) * n;
  c->last_M = -1;
  for (int p = 0; p <= maximum_iter; p++) {
    if (int rc = enqueue_pass_dev(c, c->d_sums, c->d_sums + ns_)) return rc;
    g.last = p == maximum_iter ? 1 : 0;
    hipLaunchKernelGGL(k_ieskf_step, dim3(1), dim3(ST_BLK), lds_bytes, c->stream, g);
  }
  MALIO_HIP(hipGetLastError());
  c->dev_update_pending = true;
  return MALIO_OK;
}

int ieskf_update_device_end(Ctx *c, malio_state_t *xio, double *Pio, int *stats) {
  if (!c->dev_update_pending) {
    c->err = "malio_update_iterated_end: no update was begun";
    return MALIO_ERR_BAD_ARG;
  }
  c->dev_update_pending = false;
  const int L = c->prm.lid_num, n = 17 + 6 * L;
  MALIO_HIP(hipStreamSynchronize(c->stream));
  c->stage_pending = false;
  // ---- results ----
  const DevLoop *o = reinterpret_cast<const DevLoop *>(c->h_loop_out);
  c->mm_parity = o->mm_parity;
  c->last_pass_search = o->last_search != 0;
  if (o->status == MALIO_SMALL_M_FALLBACK) {
    c->last_M = -1;  // the pass that ran folded the previous results already; the host loop starts this scan's update over
    return MALIO_SMALL_M_FALLBACK;
  }
  c->last_M = o->lastM;
  if (o->status < 0) {
    c->err = "malio_update_iterated: singular covariance in the device loop";
    return o->status;
  }
  *xio = o->x;
  if (o->valid_any) memcpy(Pio, c->h_loop_out + OUT_P_OFF, sizeof(double) * (size_t)n * n);
  if (stats) stats[0] = o->passes, stats[1] = o->searches, stats[2] = o->lastM_valid, stats[3] = o->t;
  return MALIO_OK;
}


int ieskf_update_device(Ctx *c, malio_state_t *xio, double *Pio, int *stats) {
  if (int rc = ieskf_update_device_begin(c, xio, Pio)) return rc;
  return ieskf_update_device_end(c, xio, Pio, stats);
}

}  // namespace malio

namespace malio {
// ---- gated loop: the chain of passes enqueued up front, the n x n algebra on the calling thread --------------------------
// MALIO_UPDATE_GATED. The host-driven loop pays, per pass, a stream synchronisation, the host algebra and the launch of
// the next pass' kernels before the GPU has anything to do again (~12 us of round trip around ~18 us of algebra). Here
// every pass of the loop is already in the queue; between two passes sits a gate - the last workgroup of the pass' last
// kernel (k_final_reduce, a ticket counter) - that (1) tells the host, through a sequence word in pinned memory, that the
// pass' sums are complete (they are stored by the kernels straight into pinned memory), (2) polls a second word until the
// host has published the control block of the next pass (state in the forms the kernels read, converge flag, parities,
// or `done`), and (3) copies that block into the DevLoop the pass kernels read. GPU -> host costs one PCIe latency
// instead of a completion signal; host -> GPU is a posted write: under a large BAR the block and its word live in
// (fine-grained) device memory, which the host stores into directly and the gate polls locally (-2 us per pass against
// polling pinned memory across PCIe). The first pass needs no gate: the state it starts from is known when the update
// is called, so it is launched the way the host-driven loop launches a pass (arguments by value), with gate 1 riding on
// it. A gate gives up after GATE_TIMEOUT_US (the host died or returned): the chain then drains as on `done`.

int ensure_gate_buffers(Ctx *c) {
  const size_t hdr = loop_block_doubles();
  if (!c->d_gate_ticket) {
    MALIO_HIP(hipMalloc(&c->d_gate_ticket, 256));
    MALIO_HIP(hipMemsetAsync(c->d_gate_ticket, 0, 256, c->stream));
  }
  if (!c->h_gate) {
    MALIO_HIP(hipHostMalloc((void **)&c->h_gate, sizeof(double) * hdr + 256, hipHostMallocMapped | hipHostMallocCoherent));
    MALIO_HIP(hipHostGetDevicePointer((void **)&c->d_gate, c->h_gate, 0));
    memset(c->h_gate, 0, sizeof(double) * hdr + 256);
    c->gate_stage.assign(hdr, 0.0);
  }
  // host -> GPU direction in device memory when the CPU can store there (MALIO_OPT_GATE_PINNED keeps it in pinned memory;
  // the option may change between updates: no chain is in flight then)
  if (c->opt_gate_pinned && c->d_cmd) {
    MALIO_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_cmd);
    c->d_cmd = nullptr, c->d_cmd_tried = false;
  }
  if (!c->opt_gate_pinned && !c->d_cmd && !c->d_cmd_tried) {
    c->d_cmd_tried = true;
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, c->device) == hipSuccess && large_bar) {
      if (hipExtMallocWithFlags((void **)&c->d_cmd, sizeof(double) * hdr + 256, hipDeviceMallocFinegrained) == hipSuccess) {
        MALIO_HIP(hipMemset(c->d_cmd, 0, sizeof(double) * hdr + 256));
        MALIO_HIP(hipDeviceSynchronize());
      } else {
        (void)hipGetLastError();
        c->d_cmd = nullptr;
      }
    }
  }
  return MALIO_OK;
}
int gate_words(Ctx *c, volatile int **host_msg, int **dev_msg) {  // the GPU -> host sequence word
  if (int rc = ensure_gate_buffers(c)) return rc;
  const size_t off = sizeof(double) * loop_block_doubles() + 128;
  *host_msg = reinterpret_cast<volatile int *>(c->h_gate + off);
  *dev_msg = reinterpret_cast<int *>(c->d_gate + off);
  return MALIO_OK;
}

// xchg != nullptr: this handle is one shard of a node (malio_update_iterated_node). Pass 0 is then malio_measure_node's
// (its rows need the GLOBAL extrema, which nobody has before the first exchange of a scan); from pass 1 on the chain is the
// same as on one GPU - every unit a speculating k_pass on the node's extrema of the pass before - and what the loop does
// between "sums seen" and "published" gains one step: the [sums | extrema] rows of all shards meet in host memory
// (malio_xchg_reduce), every shard forms the same sums in rank order and takes the same decision - hit, or the same pass
// again with the extrema the exchange found.
int ieskf_update_gated(Ctx *c, malio_xchg_t xchg, malio_state_t *xio, double *Pio, int *stats, double *solve_time) {
  const int L = c->prm.lid_num, n = 17 + 6 * L, maximum_iter = c->prm.max_iteration;
  const double limit = c->prm.limit > 0 ? c->prm.limit : 0.001;
  if (int rc = prepare_scan_dev(c, xio)) return rc;
  malio_measure_out_t mo0;
  int rc_mo0 = MALIO_OK;
  if (xchg) {  // before the chain's sequence numbers are drawn: malio_measure_node announces its stages through the same word
    memset(&mo0, 0, sizeof(mo0));
    rc_mo0 = malio_measure_node((malio_handle_t)c, xchg, xio, 1, &mo0, nullptr);
    if (rc_mo0 < 0) return rc_mo0;
    if (!c->scan_sorted || c->seg_pending || !c->d_tiles) {
      c->err = "malio_update_iterated_node: the scan is not in the form the one-kernel pass reads";
      return MALIO_ERR_BAD_ARG;
    }
  }
  const size_t hdr = loop_block_doubles(), nn = (size_t)DEV_NMAX * DEV_NMAX;
  if (!c->d_loopbuf) {
    MALIO_HIP(hipMalloc(&c->d_loopbuf, sizeof(double) * (hdr + 2 * nn)));
    c->d_loop = reinterpret_cast<DevLoop *>(c->d_loopbuf);
    MALIO_HIP(hipHostMalloc((void **)&c->h_loop_in, sizeof(double) * (hdr + nn), hipHostMallocMapped));
    MALIO_HIP(hipHostMalloc((void **)&c->h_loop_out, OUT_P_OFF + sizeof(double) * nn, hipHostMallocMapped | hipHostMallocCoherent));
    MALIO_HIP(hipHostGetDevicePointer((void **)&c->d_loop_out, c->h_loop_out, 0));
  }
  if (int rcg = ensure_gate_buffers(c)) return rcg;
  // layout (pinned, and device memory under a large BAR): [DevLoop block | cmd_seq (int) ... msg_seq (int at +128, pinned only)]
  DevLoop *blk = reinterpret_cast<DevLoop *>(c->gate_stage.data());
  char *cmd_home = c->d_cmd ? c->d_cmd : c->h_gate;  // where the gate reads the block and its sequence word
  volatile int *cmd_seq = reinterpret_cast<volatile int *>(cmd_home + sizeof(double) * hdr);
  volatile int *msg_seq = reinterpret_cast<volatile int *>(c->h_gate + sizeof(double) * hdr + 128);
  // The chain is made of UNITS: unit 0 is pass 0 with its arguments by value, every later unit is one pass reading the
  // control block (one kernel - k_pass, speculating on the extrema - where that is possible, else the four kernels of
  // enqueue_pass_dev), with a gate riding on its last workgroup. Unit u announces base + u + 1 and then waits for
  // base + u + 2, the block of unit u + 1. A unit is normally the next pass of the loop; after a one-kernel pass whose
  // guess of the extrema turned out wrong it is the SAME pass again (same state, neighbours and planes kept, rows
  // weighted with the now known extrema), which the loop does not count.
  const int max_units = 3 * (maximum_iter + 1) + 2;  // a pass, and up to two repeats of it (the attempt loop below)
  const int base = c->gate_epoch;
  c->gate_epoch += max_units + 4;
  if (c->gate_epoch > (1 << 30)) c->gate_epoch = 1;
  const int ns_ = sums_len(c);

  malio_state_t x_ = *xio;
  const malio_state_t x_prop = x_;
  std::vector<double> P_prop;  // (copied once pass 0 is on its way)
  std::vector<double> P_work;  // what the iterations leave; reaches the caller's P on the success path only (an early
  bool P_written = false;      // MALIO_SMALL_M_FALLBACK or an error returns x and P as they came in)
  int converge = 1, t = 0, passes = 0, searches = 0, lastM = 0;
  bool done = false;
  double solve = 0;
  auto publish = [&](int u, bool stop, bool redo) {  // the control block unit u runs on (the gate of unit u - 1 waits for it)
    memset(blk, 0, sizeof(DevLoop));
    blk->done = stop ? 1 : 0;
    if (!stop) {
      blk->converge = redo ? 0 : converge;
      blk->L = L, blk->maximum_iter = maximum_iter, blk->extrinsic_est_en = c->prm.extrinsic_est_en;
      c->mm_parity ^= 1;
      if (blk->converge) blk->search_skip = search_skip_begin(c);
      blk->mm_parity = c->mm_parity;
      // (a repeated pass folds nothing: the pass it repeats has consumed the pending fold, and its own results are the
      // repeat's results)
      blk->commit_prev = (!redo && c->last_M > 0) ? 1 : 0;
      c->last_M = -1;
      // (a repeat runs in reuse form at the SAME state: what malio_scan_get reads of the repeated search pass - world4,
      // neighbours - is untouched by it, so the pass it repeats keeps deciding where feats_down_world is read from)
      if (!redo) c->last_pass_search = blk->converge != 0;
      memcpy(blk->mm_guess, c->mm_guess, sizeof(blk->mm_guess));
      if (c->fuse_debug_bad_guess && !redo) blk->mm_guess[0] += 1.0;
      memcpy(c->fuse_guess_used, blk->mm_guess, sizeof(blk->mm_guess));
      fill_quat_const(c, &x_, blk->qc);
      fill_pass_const(c, &x_, blk->pc);
    }
    // the block in one piece, then - behind a store fence: the BAR mapping is write-combining - its sequence word
    // (only what the pass kernels read: the control words and the two forms of the state - x, x_prop and the rest of the
    // block belong to the device-resident loop)
    memcpy(cmd_home, blk, offsetof(DevLoop, tcq));
    memcpy(cmd_home + offsetof(DevLoop, qc), reinterpret_cast<const char *>(blk) + offsetof(DevLoop, qc),
           offsetof(DevLoop, loc_thresh_min) - offsetof(DevLoop, qc));
    _mm_sfence();
    __atomic_store_n(const_cast<int *>(cmd_seq), base + u + 1, __ATOMIC_RELEASE);
    _mm_sfence();
  };
  GateArgs g;
  const char *cmd_dev = c->d_cmd ? c->d_cmd : c->d_gate;
  g.dl = c->d_loop, g.cmd = reinterpret_cast<const double *>(cmd_dev);
  g.cmd_seq = reinterpret_cast<const int *>(cmd_dev + sizeof(double) * hdr);
  g.msg_seq = reinterpret_cast<int *>(c->d_gate + sizeof(double) * hdr + 128);
  g.ndoubles = (int)hdr;
  g.ticket = c->d_gate_ticket;
  g.timeout_ticks = c->gate_timeout_ticks;
  std::vector<char> unit_fused((size_t)max_units + 1, 0);
  int u_enq = xchg ? 0 : 1;  // units enqueued so far (one GPU: unit 0 below; a shard: its unit 0 is pass 1, launched after pass 0's algebra)
  auto enqueue_unit = [&]() -> int {
    const int u = u_enq;
    if (u >= max_units) {
      c->err = "malio_update_iterated: the gated loop ran out of units (the one-kernel pass keeps missing its own extrema)";
      return MALIO_ERR_BAD_ARG;
    }
    g.publish = base + u + 1, g.wait_for = base + u + 2;
    // decided when the unit is enqueued, one pass ahead (the guess itself travels in the block)
    // (a shard's unit is always the one-kernel pass: the three-kernel unit weights its rows with the extrema it finds
    // itself, which on a shard are not the node's)
    const bool fused = xchg ? true : fuse_eligible(c, /*converge: a search pass may come*/ 1, /*need_guess*/ false);
    unit_fused[u] = fused ? 1 : 0;
    u_enq++;
    return fused ? enqueue_pass_fused_dev(c, &g) : enqueue_pass_dev(c, c->d_res, c->d_res + ns_, &g);
  };
  g.publish = base + 1, g.wait_for = base + 2;
  if (!xchg) {
    if (int rc = pass_stage1(c, &x_, 1, nullptr)) return rc;
    if (int rc = pass_stage2(c, nullptr, c->d_res + ns_, c->d_res, false, &g)) return rc;
  }
  P_prop.assign(Pio, Pio + (size_t)n * n);
  // ---- the loop (esekfom.hpp:509) ----
  int rc_out = MALIO_OK;
  malio_measure_out_t mo;
  StepPre pre;
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double *tr = c->gate_trace;
  int ntr = 0;
  const double t_begin = now_us();
  int u = 0;  // the unit whose announcement the loop waits for next
  // wait for unit `uu` to announce its sums; MALIO_SMALL_M_FALLBACK when the chain drained without it (gate timeout)
  auto wait_unit = [&](int uu) -> int {
    long long spins = 0;
    while (__atomic_load_n(const_cast<int *>(msg_seq), __ATOMIC_ACQUIRE) != base + uu + 1) {
      if ((++spins & 0xFFFF) == 0 && hipStreamQuery(c->stream) != hipErrorNotReady) {
        if (__atomic_load_n(const_cast<int *>(msg_seq), __ATOMIC_ACQUIRE) == base + uu + 1) break;
        // The queue drained without this pass' word: a gate gave up waiting for this thread (descheduled, stopped in a
        // debugger: gate_body's timeout) and the rest of the chain returned at once - or the stream failed. The first
        // is no reason to fail the filter update: x and P are still the caller's, the host-driven loop redoes it.
        MALIO_HIP(hipStreamSynchronize(c->stream));  // (a failed stream surfaces here)
        if (int rcr = reset_pass_state(c)) return rcr;
        c->gate_timeouts++;
        return MALIO_SMALL_M_FALLBACK;
      }
      __builtin_ia32_pause();
    }
    return MALIO_OK;
  };
  // A shard-local failure BETWEEN two exchanges (a launch that failed, a chain out of units): the peers are on their way into
  // the next exchange and would wait there for this shard until the exchange times out (60 s), their gates polling on the GPU
  // meanwhile. This shard therefore still goes to that exchange, with the extremum no pass produces (the row a gate time-out
  // sends): every shard leaves the gated loop in step; only this one reports an error. (Failures that follow FROM an
  // exchange - finish_host, the n x n algebra, too few points - are functions of sums every shard holds identically: all
  // shards take them together, nobody is left waiting.)
  auto poison_next_exchange = [&]() {
    if (!xchg) return;
    double E[4];
    c->h_res[ns_] = INFINITY;
    (void)malio_xchg_reduce(xchg, c->h_res, ns_, c->fuse_guess_used, c->h_res, E, 60.0);
    c->node_guess_valid = false, c->node_uploaded_valid = false;
  };
  for (int i = -1; i < maximum_iter && !done; i++) {
    const int p = i + 1;
    searches += converge ? 1 : 0;
    if (ntr + 5 <= 60) tr[ntr++] = now_us() - t_begin;
    // the unit of THIS pass (missing only after a repeated pass used up the one that was enqueued ahead), then the unit
    // of pass p + 1, one ahead of the GPU
    const bool shard_first = xchg && p == 0;  // (a shard's pass 0 has run already: malio_measure_node above)
    while (!shard_first && rc_out == MALIO_OK && (u_enq <= u || (p + 1 <= maximum_iter && u_enq <= u + 1))) rc_out = enqueue_unit();
    if (rc_out != MALIO_OK) {
      passes = p + 1;
      poison_next_exchange();  // (the peers are heading for this pass' exchange)
      break;
    }
    auto t0 = std::chrono::steady_clock::now();
    if (ntr + 4 <= 60) tr[ntr++] = now_us() - t_begin;
    ieskf_step_pre(L, &x_, &x_prop, P_prop.data(), pre, true);  // :526-572 + the first inversion, under the GPU's pass
    solve += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (ntr + 3 <= 60) tr[ntr++] = now_us() - t_begin;
    const double *res = nullptr;
    for (int attempt = 0; !shard_first; attempt++) {  // the pass, and - rarely - its repeat with the right extrema
      const int rcw = wait_unit(u);
      if (rcw && !xchg) return rcw;
      res = c->h_res;
      bool hit = false;
      if (xchg) {
        // the shards' rows meet. A shard whose chain is gone (its gate gave up on this thread, or its stream failed) still
        // comes, with an extremum no pass produces: every shard then leaves the gated loop the same way.
        if (rcw) c->h_res[ns_] = INFINITY;
        double E[4];
        const int rcx = malio_xchg_reduce(xchg, c->h_res, ns_, c->fuse_guess_used, c->h_res, E, 60.0);
        if (rcx < 0 || rcw < 0 || E[0] == INFINITY) {
          if (!rcw) {  // this shard's chain is still there: let it drain
            publish(u + 1, true, false);
            if (int rcr = reset_pass_state(c)) return rcr;
          }
          c->node_guess_valid = false, c->node_uploaded_valid = false;
          return rcw < 0 ? rcw : rcx < 0 ? rcx : MALIO_SMALL_M_FALLBACK;
        }
        hit = rcx == MALIO_OK;
        memcpy(c->h_res + ns_, E, sizeof(E));  // the node's extrema where the one-GPU loop finds its own
        memcpy(c->node_guess, E, sizeof(E));
        c->node_guess_valid = true;
        fuse_note(c, hit);
        if (hit) c->node_hits++;
        else c->node_misses++;
      } else {
        if (!unit_fused[u]) break;
        fused_collect(c, nullptr, &hit);
      }
      if (hit) break;
      if (attempt >= 2) {
        c->err = "malio_update_iterated: the one-kernel pass keeps missing its own extrema";
        rc_out = MALIO_ERR_HIP;
        break;
      }
      // the guess missed: the next unit repeats this pass (reuse form, same state) with the extrema this one found
      memcpy(c->mm_guess, res + ns_, sizeof(double) * 4);
      c->mm_guess_valid = true;
      if (u_enq <= u + 1)
        if (int rc = enqueue_unit()) {
          rc_out = rc;
          poison_next_exchange();  // (the peers took the same miss and meet again for the repeat)
          break;
        }
      publish(u + 1, false, true);
      u++;
    }
    if (rc_out != MALIO_OK) {
      passes = p + 1;
      break;
    }
    passes++;
    if (ntr + 2 <= 60) tr[ntr++] = now_us() - t_begin;
    int rc;
    if (shard_first) {
      mo = mo0, rc = rc_mo0;  // (malio_measure_node has done the bookkeeping of its pass)
      memcpy(c->mm_guess, c->node_guess, sizeof(double) * 4);
    } else {
      u++;
      memset(&mo, 0, sizeof(mo));
      rc = finish_host(c, res, res + ns_, &mo);
      c->last_M = mo.M;
      memcpy(c->mm_guess, res + ns_, sizeof(double) * 4);
    }
    c->mm_guess_valid = true;
    if (rc < 0) {
      rc_out = rc;
      break;
    }
    if (mo.valid) {
      lastM = mo.M;
      if (n > mo.M) {  // esekfom.hpp:574-582 works on rows: the host-driven loop redoes this update
        rc_out = MALIO_SMALL_M_FALLBACK;
        break;
      }
      t0 = std::chrono::steady_clock::now();
      int dn = 0;
      if (!P_written) P_work.resize((size_t)n * n);
      rc = ieskf_step_post(L, maximum_iter, limit, i, &x_, &x_prop, pre, mo.HtRinvH, mo.HtRinvh, &t, &converge, &dn, P_work.data());
      solve += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (rc != MALIO_OK) {
        rc_out = rc;
        break;
      }
      P_written = true;
      done = dn != 0;
    }
    if (c->gate_debug_stall_ms > 0 && p == 1) usleep(1000 * (useconds_t)c->gate_debug_stall_ms);
    if (!done && i + 1 < maximum_iter) {
      if (shard_first) {  // the shard's unit 0: pass 1 with its arguments by value, a gate on its last workgroup
        g.publish = base + 1, g.wait_for = base + 2;
        unit_fused[0] = 1, u_enq = 1;
        if (int rcp = pass_fused(c, &x_, converge, &g, nullptr)) {
          rc_out = rcp;
          poison_next_exchange();  // (the peers launched their pass 1 and wait for its rows)
          break;
        }
      } else {
        publish(u, false, false);
      }
    }
    if (ntr + 1 <= 60) tr[ntr++] = now_us() - t_begin;
  }
  c->gate_trace_n = ntr;
  publish(u, true, false);  // the gate of the last unit that ran: everything behind it drains
  if (rc_out == MALIO_SMALL_M_FALLBACK) {
    c->last_M = -1;
    return rc_out;
  }
  if (rc_out != MALIO_OK) return rc_out;
  *xio = x_;
  if (P_written) memcpy(Pio, P_work.data(), sizeof(double) * (size_t)n * n);
  if (stats) stats[0] = passes, stats[1] = searches, stats[2] = lastM, stats[3] = t;
  if (solve_time) *solve_time += solve;
  return MALIO_OK;
}
}  // namespace malio

// developer aid (tools/probe_devloop.py): phase stamps of the last step kernel of the last device-loop update
extern "C" int malio_debug_gate_trace(malio_handle_t h, double *out60) {  // per pass: loop top, launches done, pre done, sums seen, published [us]
  if (!h || !out60) return -1;
  memcpy(out60, h->gate_trace, sizeof(double) * 60);
  return h->gate_trace_n;
}
extern "C" int malio_debug_loop_stamps(malio_handle_t h, long long *out16) {
  if (!h || !out16 || !h->h_loop_out) return MALIO_ERR_BAD_ARG;
  memcpy(out16, reinterpret_cast<const malio::DevLoop *>(h->h_loop_out)->stamps, sizeof(long long) * 16);
  return MALIO_OK;
}
