// IMU propagation primitive: ONE step of esekf::predict
// (/root/reference/MA_LIO/include/IKFoM_toolkit/esekfom/esekfom.hpp:388-492; predict_cont :171-279 and back_predict
// :281-385 are the same body run on x_cont/P_unc_ and x_unc/P_unc_) with MA-LIO's process model
// (/root/reference/MA_LIO/src/use-ikfom.hpp:67-112: get_f, df_dx, df_dw) for the runtime-parametric state.
// The loops that call it (IMU_Processing.hpp:262-400: forward propagation, the 100 Hz continuous track, the
// backward uncertainty track) stay with the caller - they are control flow over IMU messages, not data-parallel work.
//
// The reference forms dense n x n products. The Jacobians have four non-trivial 3-row bands, so here
//   F = D + dt * S,   D = I except the 2 x 2 gravity block,   S = { pos<-vel, rot<-bg, vel<-rot, vel<-ba, vel<-grav }
// is applied band by band: P <- F P F^T costs O(n^2) instead of O(n^3), and G Q G^T touches a 12 x 12 footprint.
// What is kept from the reference, because it changes values: exp(seg, scalar(1/2)) with INTEGER 1/2 == 0 makes the
// SO3 blocks of F_x1 the identity (esekfom.hpp:421,450), A_matrix(-f dt) multiplies the SO3 rows (:432-438), and the
// gravity block is Nx(g) * Mx(g, 0) (:452-464), which is the identity only up to rounding.
#include <cmath>
#include <cstring>
#include <vector>
#include "../csrc/malio_internal.hpp"
#include "manifold.hpp"

namespace malio {
using namespace mf;
namespace {

// One 3 x w band of S: rows [r, r+3) of F*P gain dt * B * P[c .. c+w)
struct Band {
  int r, c, w;
  double B[3][3];
};

}  // namespace

int predict_step(int L, malio_state_t *x, double *P, double dt, const double *Q, const double *acc,
                 const double *gyro) {
  const int n = 17 + 6 * L;
  const int i_rot = 3, i_vel = 6 * (L + 1), i_bg = i_vel + 3, i_ba = i_vel + 6, i_grav = i_vel + 9;

  // ---- process model at the PRIOR state (use-ikfom.hpp:67-112) --------------------------------------------------
  double R[3][3], a_b[3], omega[3], a_w[3];
  quat_R(x->rot, R);
  for (int k = 0; k < 3; k++) a_b[k] = acc[k] - x->ba[k], omega[k] = gyro[k] - x->bg[k];
  {
    // vel' = rot * (acc - ba) + grav. The reference rotates with the quaternion product (Eigen's
    // Quaternion * Vector3: v + w t + q x t, t = 2 q x v); do the same rather than going through R.
    const double *q = x->rot;
    double t[3] = {2 * (q[1] * a_b[2] - q[2] * a_b[1]), 2 * (q[2] * a_b[0] - q[0] * a_b[2]),
                   2 * (q[0] * a_b[1] - q[1] * a_b[0])};
    a_w[0] = a_b[0] + q[3] * t[0] + (q[1] * t[2] - q[2] * t[1]);
    a_w[1] = a_b[1] + q[3] * t[1] + (q[2] * t[0] - q[0] * t[2]);
    a_w[2] = a_b[2] + q[3] * t[2] + (q[0] * t[1] - q[1] * t[0]);
  }
  double Bg[3][2], Hg[3][3];
  s2_Bx(x->grav, Bg);
  hat3(x->grav, Hg);

  Band bands[5];
  int nb = 0;
  {  // pos <- vel: I
    Band &b = bands[nb++];
    b.r = 0, b.c = i_vel, b.w = 3;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) b.B[i][j] = i == j;
  }
  double A[3][3];
  {  // rot <- bg: A_matrix(-omega dt) * (-I)    (esekfom.hpp:418-438 on df_dx (3, i_bg) = -I)
    double seg[3] = {-1 * omega[0] * dt, -1 * omega[1] * dt, -1 * omega[2] * dt};
    A_matrix(seg, A);
    Band &b = bands[nb++];
    b.r = i_rot, b.c = i_bg, b.w = 3;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) b.B[i][j] = -A[i][j];
  }
  {  // vel <- rot: -R hat(acc - ba)
    double Ha[3][3];
    hat3(a_b, Ha);
    Band &b = bands[nb++];
    b.r = i_vel, b.c = i_rot, b.w = 3;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) b.B[i][j] = -(R[i][0] * Ha[0][j] + R[i][1] * Ha[1][j] + R[i][2] * Ha[2][j]);
  }
  {  // vel <- ba: -R
    Band &b = bands[nb++];
    b.r = i_vel, b.c = i_ba, b.w = 3;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) b.B[i][j] = -R[i][j];
  }
  {  // vel <- grav: S2_Mx(0) = -hat(g) Bx   (S2.hpp:276-283)
    Band &b = bands[nb++];
    b.r = i_vel, b.c = i_grav, b.w = 2;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 2; j++) b.B[i][j] = -(Hg[i][0] * Bg[0][j] + Hg[i][1] * Bg[1][j] + Hg[i][2] * Bg[2][j]);
  }
  // gravity block of F_x1: Nx(g) * Mx(g, 0), Nx = Bx^T hat(g) / |g|^2 (S2.hpp:269-274). g is unchanged by oplus
  // because its flow is zero (exp(0) == identity exactly).
  double Sg[2][2];
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) {
        double nx = (Bg[0][i] * Hg[0][k] + Bg[1][i] * Hg[1][k] + Bg[2][i] * Hg[2][k]) / G_LEN / G_LEN;
        s += nx * bands[4].B[k][j];
      }
      Sg[i][j] = s;
    }

  // ---- x.oplus(f, dt) (esekfom.hpp:398; vect.hpp, SOn.hpp:250-253) -------------------------------------------------
  {
    double vel0[3] = {x->vel[0], x->vel[1], x->vel[2]};
    for (int k = 0; k < 3; k++) x->pos[k] += dt * vel0[k];
    double h = dt / 2, c, s, dq[4];
    cos_sinc(h * h * (omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]), c, s);
    dq[0] = s * h * omega[0], dq[1] = s * h * omega[1], dq[2] = s * h * omega[2], dq[3] = c;
    qmul(x->rot, dq, x->rot);
    for (int k = 0; k < 3; k++) x->vel[k] += dt * (a_w[k] + x->grav[k]);
    // offset_R, offset_T, bg, ba, grav: zero flow
  }

  if (!P) return MALIO_OK;
  // ---- T = F P ------------------------------------------------------------------------------------------------------
  std::vector<double> T(P, P + (size_t)n * n);
  for (int k = 0; k < nb; k++) {
    const Band &b = bands[k];
    for (int i = 0; i < 3; i++) {
      double *t = &T[(size_t)(b.r + i) * n];
      for (int j = 0; j < b.w; j++) {
        const double f = dt * b.B[i][j];
        const double *p = P + (size_t)(b.c + j) * n;
        for (int c = 0; c < n; c++) t[c] += f * p[c];
      }
    }
  }
  {  // gravity rows: D block (these rows have no band)
    double r0, r1;
    for (int c = 0; c < n; c++) {
      r0 = Sg[0][0] * P[(size_t)i_grav * n + c] + Sg[0][1] * P[(size_t)(i_grav + 1) * n + c];
      r1 = Sg[1][0] * P[(size_t)i_grav * n + c] + Sg[1][1] * P[(size_t)(i_grav + 1) * n + c];
      T[(size_t)i_grav * n + c] = r0, T[(size_t)(i_grav + 1) * n + c] = r1;
    }
  }
  // ---- P = T F^T: column c of the result = T * F[c, :]^T -----------------------------------------------------------
  for (int r = 0; r < n; r++) {
    const double *t = &T[(size_t)r * n];
    double *p = P + (size_t)r * n;
    std::memcpy(p, t, sizeof(double) * n);
    p[i_grav] = Sg[0][0] * t[i_grav] + Sg[0][1] * t[i_grav + 1];
    p[i_grav + 1] = Sg[1][0] * t[i_grav] + Sg[1][1] * t[i_grav + 1];
    for (int k = 0; k < nb; k++) {
      const Band &b = bands[k];
      for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int j = 0; j < b.w; j++) s += b.B[i][j] * t[b.c + j];
        p[b.r + i] += dt * s;
      }
    }
  }
  // ---- + (dt f_w) Q (dt f_w)^T: non-zero rows of f_w_final are rot (-A on ng), vel (-R on na), bg (I on nbg),
  //      ba (I on nba) (use-ikfom.hpp:104-112, esekfom.hpp:436-438) ------------------------------------------------------
  double G[12][12];
  std::memset(G, 0, sizeof(G));
  const int rows[4] = {i_rot, i_vel, i_bg, i_ba};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      G[i][j] = dt * -A[i][j];
      G[3 + i][3 + j] = dt * -R[i][j];
    }
  for (int i = 0; i < 3; i++) G[6 + i][6 + i] = dt, G[9 + i][9 + i] = dt;
  double GQ[12][12];
  for (int i = 0; i < 12; i++)
    for (int j = 0; j < 12; j++) {
      double s = 0;
      for (int k = 0; k < 12; k++) s += G[i][k] * Q[k * 12 + j];
      GQ[i][j] = s;
    }
  for (int i = 0; i < 12; i++)
    for (int j = 0; j < 12; j++) {
      double s = 0;
      for (int k = 0; k < 12; k++) s += GQ[i][k] * G[j][k];
      P[(size_t)(rows[i / 3] + i % 3) * n + rows[j / 3] + j % 3] += s;
    }
  return MALIO_OK;
}

}  // namespace malio
