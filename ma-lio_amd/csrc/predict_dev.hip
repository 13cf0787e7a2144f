// Row f-3 on the device: chains of esekf::predict steps (esekfom.hpp:388-492; predict_cont :171-279 and back_predict
// :281-385 are the same body on other (state, covariance) pairs) with MA-LIO's process model (use-ikfom.hpp:67-112).
//
// A chain is serial - step k + 1 needs the state and covariance of step k - so one chain is ONE workgroup; what the
// device can batch is what the reference runs one after the other on its main thread: the three tracks of a scan
// (kf.predict on (x_, P_), predict_cont on (x_cont, P_unc_), back_predict on (x_unc, P_unc_): IMU_Processing.hpp:332,345,
// 364,386,399) are independent of each other, so they run as three workgroups side by side, state and covariance in LDS
// from the first step to the last, the state after every step stored for whoever builds the spline knots and the pose
// tables from it. Same arithmetic as host/predict.cpp (banded F P F^T, the shared host/manifold.hpp functions compiled for
// the device, every entry's terms added in the same order); what differs is libm (sin / cos of the device library).
#include <vector>
#include "malio_internal.hpp"
#include "../host/manifold.hpp"

namespace malio {
using namespace mf;

constexpr int PC_MAT = 256;           // threads of the matrix phases (4 waves)
constexpr int PC_BLK = PC_MAT + 64;   // + one wave that prepares the NEXT step meanwhile

struct ChainTrack {
  int K;        // steps
  int off;      // offset of this track's steps in dt / acc / gyro / out_states
  int has_P;
};
struct ChainArgs {
  int L, n_tracks;
  ChainTrack tr[4];
  malio_state_t *x;        // [n_tracks] in: start, out: end
  double *P;               // [n_tracks][n*n]
  const double *dt;        // [sum K]
  const double *acc;       // [sum K][3]
  const double *gyro;      // [sum K][3]
  const double *Q;         // [12][12]
  malio_state_t *out;      // [sum K] state after every step, or null
};

// what one step's matrix phases need from the process model at the step's PRIOR state
struct StepBands {
  double B[5][3][3];  // pos<-vel, rot<-bg, vel<-rot, vel<-ba, vel<-grav (3 x 2 used)
  double Sg[2][2];
  double G[12][12];   // dt * f_w footprint (rows rot, vel, bg, ba)
  double dt;
};

// what does not change along a chain (gravity has zero flow, so g, Bx(g), hat(g) stay what they are): pos<-vel, vel<-grav,
// the gravity block of F_x1, the zeros of G
__device__ void prepare_chain(const malio_state_t *x, StepBands &sb) {
  double Bg[3][2], Hg[3][3];
  s2_Bx(x->grav, Bg);
  hat3(x->grav, Hg);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) sb.B[0][i][j] = i == j;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 2; j++) sb.B[4][i][j] = -(Hg[i][0] * Bg[0][j] + Hg[i][1] * Bg[1][j] + Hg[i][2] * Bg[2][j]);
    sb.B[4][i][2] = 0;
  }
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) {
        double nx = (Bg[0][i] * Hg[0][k] + Bg[1][i] * Hg[1][k] + Bg[2][i] * Hg[2][k]) / G_LEN / G_LEN;
        s += nx * sb.B[4][k][j];
      }
      sb.Sg[i][j] = s;
    }
  for (int i = 0; i < 12; i++)
    for (int j = 0; j < 12; j++) sb.G[i][j] = 0;
}
// process model, state-dependent bands of F and x.oplus(f, dt) of one step: host/predict.cpp line by line (one lane)
__device__ void prepare_step(malio_state_t *x, double dt, const double *acc, const double *gyro, StepBands &sb) {
  double R[3][3], a_b[3], omega[3], a_w[3];
  quat_R(x->rot, R);
  for (int k = 0; k < 3; k++) a_b[k] = acc[k] - x->ba[k], omega[k] = gyro[k] - x->bg[k];
  {
    const double *q = x->rot;
    double t[3] = {2 * (q[1] * a_b[2] - q[2] * a_b[1]), 2 * (q[2] * a_b[0] - q[0] * a_b[2]),
                   2 * (q[0] * a_b[1] - q[1] * a_b[0])};
    a_w[0] = a_b[0] + q[3] * t[0] + (q[1] * t[2] - q[2] * t[1]);
    a_w[1] = a_b[1] + q[3] * t[1] + (q[2] * t[0] - q[0] * t[2]);
    a_w[2] = a_b[2] + q[3] * t[2] + (q[0] * t[1] - q[1] * t[0]);
  }
  double A[3][3];
  {
    double seg[3] = {-1 * omega[0] * dt, -1 * omega[1] * dt, -1 * omega[2] * dt};
    A_matrix(seg, A);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) sb.B[1][i][j] = -A[i][j];
  }
  {
    double Ha[3][3];
    hat3(a_b, Ha);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) sb.B[2][i][j] = -(R[i][0] * Ha[0][j] + R[i][1] * Ha[1][j] + R[i][2] * Ha[2][j]);
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) sb.B[3][i][j] = -R[i][j];
  {  // x.oplus(f, dt)
    double vel0[3] = {x->vel[0], x->vel[1], x->vel[2]};
    for (int k = 0; k < 3; k++) x->pos[k] += dt * vel0[k];
    double h = dt / 2, c, s, dq[4];
    cos_sinc(h * h * (omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]), c, s);
    dq[0] = s * h * omega[0], dq[1] = s * h * omega[1], dq[2] = s * h * omega[2], dq[3] = c;
    qmul(x->rot, dq, x->rot);
    for (int k = 0; k < 3; k++) x->vel[k] += dt * (a_w[k] + x->grav[k]);
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      sb.G[i][j] = dt * -A[i][j];
      sb.G[3 + i][3 + j] = dt * -R[i][j];
    }
  for (int i = 0; i < 3; i++) sb.G[6 + i][6 + i] = dt, sb.G[9 + i][9 + i] = dt;
  sb.dt = dt;
}

// One workgroup = one track. Wave 4 walks the state through the steps (the state's flow does not depend on P), leaves
// each step's bands and G Q in one of two buffers; waves 0-3 apply them to the covariance one step behind:
// P <- F P F^T + G Q G^T in the banded form of host/predict.cpp, every entry's terms in the same order.
// F differs from the identity in 11 rows (pos, rot, vel: bands; gravity: its 2 x 2 block), so F P differs from P in 11
// rows and (F P) F^T from F P in 11 columns: per step 11 n entries of new rows (TR), 11 n entries of new columns (PC) and
// the 72 entries of G Q G^T's footprint that lie in untouched columns (bg, ba) - not two n x n sweeps.
__global__ void __launch_bounds__(PC_BLK) k_predict_chain(ChainArgs a) {
  extern __shared__ double lds[];
  const int tix = blockIdx.x;
  if (tix >= a.n_tracks) return;
  const ChainTrack tk = a.tr[tix];
  const int L = a.L, n = 17 + 6 * L, ns = n;
  const int i_rot = 3, i_vel = 6 * (L + 1), i_grav = i_vel + 9;
  const int tid = threadIdx.x;
  const bool prep_wave = tid >= PC_MAT;
  double *P = lds;                  // [n][n]
  double *TR = P + n * ns;          // [11][n] rows pos, rot, vel, grav of F P
  double *PC = TR + 11 * ns;        // [n][11] columns pos, rot, vel, grav of (F P) F^T (+ G Q G^T)
  double *PX = PC + 11 * ns;        // [12][6] footprint rows x (bg, ba) columns: (F P)[r][c] + (G Q G^T)
  double *GQ = PX + 72;             // [144] G Q of the step being prepared
  double *GG = GQ + 144;            // [2][144] G Q G^T, per step
  double *Qs = GG + 288;            // [144]
  __shared__ malio_state_t xs;
  __shared__ StepBands sbuf[2];
  double *Pg = a.P + (size_t)tix * n * n;
  if (tid == PC_MAT) {
    xs = a.x[tix];
    prepare_chain(&xs, sbuf[0]);
    prepare_chain(&xs, sbuf[1]);
  }
  if (tk.has_P) {
    if (!prep_wave)
      for (int e = tid; e < n * n; e += PC_MAT) P[e] = Pg[e];
    if (tid < 144) Qs[tid] = a.Q[tid];
  }
  // banded index b in 0..10 -> row / column of P, and the bands that end there (first band, how many)
  auto bidx = [&](int b) { return b < 6 ? b : (b < 9 ? i_vel + (b - 6) : i_grav + (b - 9)); };
  // footprint index g in 0..11 -> row / column (rot, vel, bg, ba)
  auto gidx = [&](int g) { return g < 3 ? i_rot + g : i_vel + (g - 3); };
  auto banded_of = [&](int r) { return r < 6 ? r : (r >= i_vel && r < i_vel + 3 ? 6 + (r - i_vel) : (r >= i_grav ? 9 + (r - i_grav) : -1)); };
  auto foot_of = [&](int r) { return (r >= i_rot && r < i_rot + 3) ? r - i_rot : ((r >= i_vel && r < i_vel + 9) ? 3 + (r - i_vel) : -1); };
  const int band_c[5] = {i_vel, i_vel + 3, i_rot, i_vel + 6, i_grav}, band_w[5] = {3, 3, 3, 3, 2};  // columns: vel, bg, rot, ba, grav
  // this thread's entries of the two phases (at most two each): the index arithmetic once, not per step
  constexpr int EPT = (11 * DEV_NMAX + 72 + PC_MAT - 1) / PC_MAT;
  int rb_[EPT], rc_[EPT];              // rows phase: banded index b, column c   (b < 0: none)
  int cr_[EPT], cb_[EPT], cq_[EPT];    // columns phase: row r, banded index b (or -1), footprint pair q (or -1)
#pragma unroll
  for (int u = 0; u < EPT; u++) {
    const int e = tid + u * PC_MAT;
    rb_[u] = e < 11 * n ? e / n : -1, rc_[u] = e < 11 * n ? e - (e / n) * n : 0;
    cr_[u] = -1, cb_[u] = -1, cq_[u] = -1;
    if (e < 11 * n) cr_[u] = e / 11, cb_[u] = e - 11 * (e / 11);
    else if (e < 11 * n + 72) cq_[u] = e - 11 * n;
  }
  __syncthreads();
  // pipeline: iteration `it` prepares step it (wave 4) while the covariance takes step it - 1 (waves 0-3)
  for (int it = 0; it <= tk.K; it++) {
    const StepBands &sb = sbuf[(it - 1) & 1];
    const double *gg = GG + 144 * ((it - 1) & 1);
    const bool apply = !prep_wave && tk.has_P && it > 0;
    if (prep_wave) {
      if (tid == PC_MAT && it < tk.K) {
        const int si = tk.off + it;
        prepare_step(&xs, a.dt[si], a.acc + 3 * (size_t)si, a.gyro + 3 * (size_t)si, sbuf[it & 1]);
      }
      if (a.out && it < tk.K) {  // the state after this step, one double per lane (same wave: ordered behind lane 0's writes)
        const int w = tid - PC_MAT;
        if (w < (int)(sizeof(malio_state_t) / sizeof(double)))
          reinterpret_cast<double *>(a.out + tk.off + it)[w] = reinterpret_cast<const double *>(&xs)[w];
      }
    } else if (apply) {
      // ---- the 11 rows of T = F P that differ from P ----
      const double dt = sb.dt;
#pragma unroll
      for (int u = 0; u < EPT; u++) {
        const int b = rb_[u], c = rc_[u];
        if (b < 0) continue;
        const int r = bidx(b);
        double t;
        if (b >= 9) {  // gravity rows: the D block
          t = sb.Sg[b - 9][0] * P[i_grav * ns + c] + sb.Sg[b - 9][1] * P[(i_grav + 1) * ns + c];
        } else {
          t = P[r * ns + c];
          const int i = b % 3, k0 = b < 3 ? 0 : (b < 6 ? 1 : 2), k1 = b < 6 ? k0 + 1 : 5;
          for (int k = k0; k < k1; k++)
            for (int j = 0; j < band_w[k]; j++) t += (dt * sb.B[k][i][j]) * P[(band_c[k] + j) * ns + c];
        }
        TR[b * ns + c] = t;
      }
    }
    if (tk.has_P) __syncthreads();
    if (prep_wave && tk.has_P && it < tk.K) {  // (G Q) G^T of the step just prepared, while the other waves do the columns
      const StepBands &sn = sbuf[it & 1];
      double *gn = GG + 144 * (it & 1);
      for (int e = tid - PC_MAT; e < 144; e += 64) {
        const int i = e / 12, j = e - 12 * i;
        double s = 0;
        for (int k = 0; k < 12; k++) s += sn.G[i][k] * Qs[k * 12 + j];
        GQ[e] = s;
      }
      // (same wave: the 144 entries of G Q are complete before any lane goes on - LDS operations of a wave retire in order)
      for (int e = tid - PC_MAT; e < 144; e += 64) {
        const int i = e / 12, j = e - 12 * i;
        double s = 0;
        for (int k = 0; k < 12; k++) s += GQ[i * 12 + k] * sn.G[j][k];
        gn[e] = s;
      }
    }
    if (apply) {
      // ---- the 11 columns of P = T F^T that differ from T, with G Q G^T where it lands; row r of T is TR or P ----
      const double dt = sb.dt;
#pragma unroll
      for (int u = 0; u < EPT; u++) {
        if (cb_[u] >= 0) {
          const int r = cr_[u], b = cb_[u], c = bidx(b);
          const int rb = banded_of(r);
          const double *t = rb >= 0 ? TR + rb * ns : P + r * ns;
          double p;
          if (b >= 9) {
            p = sb.Sg[b - 9][0] * t[i_grav] + sb.Sg[b - 9][1] * t[i_grav + 1];
          } else {
            p = t[c];
            const int i = b % 3, k0 = b < 3 ? 0 : (b < 6 ? 1 : 2), k1 = b < 6 ? k0 + 1 : 5;
            for (int k = k0; k < k1; k++) {
              double s = 0;
              for (int j = 0; j < band_w[k]; j++) s += sb.B[k][i][j] * t[band_c[k] + j];
              p += dt * s;
            }
          }
          const int gi = foot_of(r), gj = foot_of(c);
          if (gi >= 0 && gj >= 0) p += gg[gi * 12 + gj];
          PC[r * 11 + b] = p;
        } else if (cq_[u] >= 0) {  // footprint rows x (bg, ba) columns: no band ends in these columns
          const int q = cq_[u], gi = q / 6, gj = 6 + (q - 6 * gi);
          const int r = gidx(gi), c = gidx(gj);
          const int rb = banded_of(r);
          const double tv = rb >= 0 ? TR[rb * ns + c] : P[r * ns + c];
          PX[q] = tv + gg[gi * 12 + gj];
        }
      }
    }
    if (tk.has_P) __syncthreads();
    if (apply) {
      // ---- write back: new rows first, then (after the barrier) the new columns and the bg / ba footprint over them ----
      for (int e = tid; e < 11 * n; e += PC_MAT) {
        const int b = e / n, c = e - b * n;
        P[bidx(b) * ns + c] = TR[e];
      }
    }
    if (tk.has_P) __syncthreads();
    if (apply) {
      for (int e = tid; e < 11 * n + 72; e += PC_MAT) {
        if (e < 11 * n) {
          const int r = e / 11, b = e - 11 * r;
          P[r * ns + bidx(b)] = PC[e];
        } else {
          const int q = e - 11 * n, gi = q / 6, gj = 6 + (q - 6 * gi);
          P[gidx(gi) * ns + gidx(gj)] = PX[q];
        }
      }
    }
    __syncthreads();  // step `it` prepared, step it - 1 applied
  }
  if (tid == PC_MAT) a.x[tix] = xs;
  if (tk.has_P && !prep_wave)
    for (int e = tid; e < n * n; e += PC_MAT) Pg[e] = P[e];
}

}  // namespace malio

using namespace malio;

extern "C" int malio_predict_chain(malio_handle_t h, int n_tracks, malio_state_t *x, double *P, const int *K, const double *dt,
                                   const double *acc, const double *gyro, const double *Q, malio_state_t *out_states) {
  if (!h || n_tracks < 1 || n_tracks > 4 || !x || !K || !dt || !acc || !gyro || (P && !Q)) return MALIO_ERR_BAD_ARG;
  Ctx *c = h;
  MALIO_HIP(hipSetDevice(c->device));
  const int L = c->prm.lid_num, n = 17 + 6 * L;
  ChainArgs a;
  memset(&a, 0, sizeof(a));
  a.L = L, a.n_tracks = n_tracks;
  int tot = 0;
  for (int t = 0; t < n_tracks; t++) {
    if (K[t] < 0) return MALIO_ERR_BAD_ARG;
    a.tr[t].K = K[t], a.tr[t].off = tot, a.tr[t].has_P = P ? 1 : 0;
    tot += K[t];
  }
  if (tot == 0) return MALIO_OK;
  // one staging block up (states, covariances, IMU samples, Q), one down (states, covariances, the per-step states)
  const size_t nn = (size_t)n * n;
  const size_t b_x = sizeof(malio_state_t) * n_tracks, b_P = P ? sizeof(double) * nn * n_tracks : 0, b_dt = sizeof(double) * tot,
               b_v = sizeof(double) * 3 * tot, b_Q = P ? sizeof(double) * 144 : 0, b_out = out_states ? sizeof(malio_state_t) * tot : 0;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_x = 0, o_P = o_x + al(b_x), o_dt = o_P + al(b_P), o_acc = o_dt + al(b_dt), o_gy = o_acc + al(b_v),
               o_Q = o_gy + al(b_v), o_out = o_Q + al(b_Q), total = o_out + al(b_out);
  char *hst = nullptr;
  if (int rcs = host_stage(c, total, (void **)&hst)) return rcs;
  ArenaScope sc(c->arena);
  char *dev = nullptr;
  MALIO_HIP(sc.get(&dev, total));
  memcpy(hst + o_x, x, b_x);
  if (P) memcpy(hst + o_P, P, b_P), memcpy(hst + o_Q, Q, b_Q);
  memcpy(hst + o_dt, dt, b_dt), memcpy(hst + o_acc, acc, b_v), memcpy(hst + o_gy, gyro, b_v);
  MALIO_HIP(hipMemcpyAsync(dev, hst, o_out, hipMemcpyHostToDevice, c->stream));
  a.x = reinterpret_cast<malio_state_t *>(dev + o_x), a.P = reinterpret_cast<double *>(dev + o_P);
  a.dt = reinterpret_cast<const double *>(dev + o_dt), a.acc = reinterpret_cast<const double *>(dev + o_acc);
  a.gyro = reinterpret_cast<const double *>(dev + o_gy), a.Q = reinterpret_cast<const double *>(dev + o_Q);
  a.out = out_states ? reinterpret_cast<malio_state_t *>(dev + o_out) : nullptr;
  const size_t lds_bytes = sizeof(double) * (nn + 22 * (size_t)n + 72 + 144 + 288 + 144);
  hipLaunchKernelGGL(k_predict_chain, dim3(n_tracks), dim3(PC_BLK), lds_bytes, c->stream, a);
  MALIO_HIP(hipGetLastError());
  MALIO_HIP(hipMemcpyAsync(hst, dev, total, hipMemcpyDeviceToHost, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  memcpy(x, hst + o_x, b_x);
  if (P) memcpy(P, hst + o_P, b_P);
  if (out_states) memcpy(out_states, hst + o_out, b_out);
  return MALIO_OK;
}
