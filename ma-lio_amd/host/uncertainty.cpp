// Host-side SE(3) covariance algebra of the reference's uncertainty tables (a15, "stays on host"):
// Barfoot's 4th-order compounding of pose covariances and the per-point 3x3 covariance, as used to build
// pose_unc[lid][k] (/root/reference/MA_LIO/src/laserMapping.cpp:1028-1048), the per-interval entries
// (src/IMU_Processing.hpp:484-494) and kf.temporal_comp (:510-522). Mirrors include/associate_uct.hpp:
//   malio_compound_pose_cov      == compoundPoseWithCov     (:85-142, method 2)
//   malio_compound_inv_pose_cov  == compoundInvPoseWithCov  (:29-83,  method 2)
//   malio_eval_point_uncertainty == evalPointUncertainty    (:153-175)
// Output may alias pose_2 exactly as the reference's call sites do; the order of reads and writes follows
// the reference, including compoundPoseWithCov taking the adjoint of pose_2.T_ AFTER pose_cp.T_ was written.
#include <cstring>
#include "../../include/malio.h"

namespace {

struct M6 {
  double a[6][6];
};
void rot_of(const double q[4], double R[3][3]) {  // Eigen toRotationMatrix, q = (x,y,z,w)
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x,
         txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0][0] = 1 - (tyy + tzz), R[0][1] = txy - twz, R[0][2] = txz + twy;
  R[1][0] = txy + twz, R[1][1] = 1 - (txx + tzz), R[1][2] = tyz - twx;
  R[2][0] = txz - twy, R[2][1] = tyz + twx, R[2][2] = 1 - (txx + tyy);
}
void qmul(const double a[4], const double b[4], double r[4]) {
  double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  r[0] = x, r[1] = y, r[2] = z, r[3] = w;
}
void qrot(const double q[4], const double v[3], double r[3]) {  // Eigen _transformVector
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  for (double &u : uv) u += u;
  double c2[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int k = 0; k < 3; k++) r[k] = v[k] + q[3] * uv[k] + c2[k];
}
void set_T(malio_pose_t &p) {
  double R[3][3];
  rot_of(p.q, R);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) p.T[i * 4 + j] = R[i][j];
    p.T[i * 4 + 3] = p.t[i];
  }
  p.T[12] = p.T[13] = p.T[14] = 0, p.T[15] = 1;
}
// adjointMatrix(T.inverse()) for a rigid T (associate_uct.hpp:8-15)
M6 adjoint_of_inverse(const double T[16]) {
  double Rt[3][3], ti[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Rt[i][j] = T[j * 4 + i];
  for (int i = 0; i < 3; i++) ti[i] = -(Rt[i][0] * T[3] + Rt[i][1] * T[7] + Rt[i][2] * T[11]);
  double sk[3][3] = {{0, -ti[2], ti[1]}, {ti[2], 0, -ti[0]}, {-ti[1], ti[0], 0}};
  M6 A;
  memset(&A, 0, sizeof(A));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      A.a[i][j] = Rt[i][j], A.a[3 + i][3 + j] = Rt[i][j];
      A.a[i][3 + j] = sk[i][0] * Rt[0][j] + sk[i][1] * Rt[1][j] + sk[i][2] * Rt[2][j];
    }
  return A;
}
M6 mul(const M6 &A, const M6 &B) {
  M6 C;
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += A.a[i][k] * B.a[k][j];
      C.a[i][j] = s;
    }
  return C;
}
M6 transpose(const M6 &A) {
  M6 C;
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) C.a[i][j] = A.a[j][i];
  return C;
}
struct M3 {
  double a[3][3];
};
M3 blk(const M6 &A, int r, int c) {
  M3 m;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) m.a[i][j] = A.a[r + i][c + j];
  return m;
}
M3 mul3(const M3 &A, const M3 &B) {
  M3 C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C.a[i][j] = A.a[i][0] * B.a[0][j] + A.a[i][1] * B.a[1][j] + A.a[i][2] * B.a[2][j];
  return C;
}
M3 add3(const M3 &A, const M3 &B) {
  M3 C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C.a[i][j] = A.a[i][j] + B.a[i][j];
  return C;
}
M3 tr3(const M3 &A) {
  M3 C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C.a[i][j] = A.a[j][i];
  return C;
}
M3 covop1(const M3 &B) {  // :17-21
  double t = B.a[0][0] + B.a[1][1] + B.a[2][2];
  M3 A = B;
  for (int i = 0; i < 3; i++) A.a[i][i] -= t;
  return A;
}
M3 covop2(const M3 &B, const M3 &C) { return add3(mul3(covop1(B), covop1(C)), covop1(mul3(C, B))); }  // :23-27
void put(M6 &A, int r, int c, const M3 &m) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A.a[r + i][c + j] = m.a[i][j];
}
// cov_1_prime + cov_2 + (A1 cov_2 + cov_2 A1^T + A2 cov_1' + cov_1' A2^T)/12 + B/4   (:53-81 == :106-134)
void fourth_order(const M6 &c1p, const M6 &c2, double out[36]) {
  M3 c1rr = blk(c1p, 0, 0), c1rp = blk(c1p, 0, 3), c1pp = blk(c1p, 3, 3);
  M3 c2rr = blk(c2, 0, 0), c2rp = blk(c2, 0, 3), c2pp = blk(c2, 3, 3);
  M6 A1, A2, B;
  memset(&A1, 0, sizeof(A1)), memset(&A2, 0, sizeof(A2)), memset(&B, 0, sizeof(B));
  put(A1, 0, 0, covop1(c1pp)), put(A1, 0, 3, covop1(add3(c1rp, tr3(c1rp)))), put(A1, 3, 3, covop1(c1pp));
  put(A2, 0, 0, covop1(c2pp)), put(A2, 0, 3, covop1(add3(c2rp, tr3(c2rp)))), put(A2, 3, 3, covop1(c2pp));
  M3 Brr = add3(add3(covop2(c1pp, c2rr), covop2(tr3(c1rp), c2rp)), add3(covop2(c1rp, tr3(c2rp)), covop2(c1rr, c2pp)));
  M3 Brp = add3(covop2(c1pp, tr3(c2rp)), covop2(tr3(c1rp), c2pp));
  M3 Bpp = covop2(c1pp, c2pp);
  put(B, 0, 0, Brr), put(B, 0, 3, Brp), put(B, 3, 0, tr3(Brp)), put(B, 3, 3, Bpp);
  M6 S1 = mul(A1, c2), S2 = mul(c2, transpose(A1)), S3 = mul(A2, c1p), S4 = mul(c1p, transpose(A2));
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++)
      out[i * 6 + j] = c1p.a[i][j] + c2.a[i][j] + (S1.a[i][j] + S2.a[i][j] + S3.a[i][j] + S4.a[i][j]) / 12 + B.a[i][j] / 4;
}
M6 load6(const double *c) {
  M6 m;
  memcpy(m.a, c, sizeof(m.a));
  return m;
}

}  // namespace

extern "C" {

int malio_compound_pose_cov(const malio_pose_t *pose_1, const malio_pose_t *pose_2, malio_pose_t *pose_cp) {
  if (!pose_1 || !pose_2 || !pose_cp) return MALIO_ERR_BAD_ARG;
  const M6 c1 = load6(pose_1->cov), c2 = load6(pose_2->cov);
  double q[4], t[3];
  qmul(pose_1->q, pose_2->q, q);            // :90
  qrot(pose_1->q, pose_2->t, t);            // :91
  for (int k = 0; k < 3; k++) t[k] += pose_1->t[k];
  memcpy(pose_cp->q, q, sizeof(q)), memcpy(pose_cp->t, t, sizeof(t));
  set_T(*pose_cp);                          // :93-97 (overwrites pose_2->T when aliased)
  M6 Ad = adjoint_of_inverse(pose_2->T);    // :99 reads pose_2.T_ AFTER that write
  M6 c1p = mul(mul(Ad, c1), transpose(Ad)); // :100
  fourth_order(c1p, c2, pose_cp->cov);      // :106-135
  return MALIO_OK;
}

int malio_compound_inv_pose_cov(const malio_pose_t *pose_1, const malio_pose_t *pose_2, malio_pose_t *pose_cp) {
  if (!pose_1 || !pose_2 || !pose_cp) return MALIO_ERR_BAD_ARG;
  const M6 c1 = load6(pose_1->cov), c2 = load6(pose_2->cov);
  const double q1c[4] = {-pose_1->q[0], -pose_1->q[1], -pose_1->q[2], pose_1->q[3]};
  double q[4], d[3], t[3];
  qmul(q1c, pose_2->q, q);                  // :35
  for (int k = 0; k < 3; k++) d[k] = pose_2->t[k] - pose_1->t[k];
  qrot(q1c, d, t);                          // :36
  memcpy(pose_cp->q, q, sizeof(q)), memcpy(pose_cp->t, t, sizeof(t));
  set_T(*pose_cp);                          // :38-42
  M6 Ad = adjoint_of_inverse(pose_cp->T);   // :44
  M6 c1p = mul(mul(Ad, c1), transpose(Ad)); // :45
  fourth_order(c1p, c2, pose_cp->cov);      // :53-81
  return MALIO_OK;
}

// cov_point (3x3 row-major) = (G blkdiag(1e4 pose.cov_, 0.1 I3) G^T)[0:3,0:3], G = [I | -[p']x | R], p' = T (0.05 p, 1)
int malio_eval_point_uncertainty(const malio_point_t *pi, const malio_pose_t *pose, double cov_point[9]) {
  if (!pi || !pose || !cov_point) return MALIO_ERR_BAD_ARG;
  const double pc[4] = {pi->x * 0.05, pi->y * 0.05, pi->z * 0.05, 1.0};
  double Tp[3];
  for (int i = 0; i < 3; i++) Tp[i] = pose->T[i * 4] * pc[0] + pose->T[i * 4 + 1] * pc[1] + pose->T[i * 4 + 2] * pc[2] + pose->T[i * 4 + 3];
  double G[3][9];
  memset(G, 0, sizeof(G));
  for (int i = 0; i < 3; i++) G[i][i] = 1.0;
  G[0][4] = Tp[2], G[0][5] = -Tp[1], G[1][3] = -Tp[2], G[1][5] = Tp[0], G[2][3] = Tp[1], G[2][4] = -Tp[0];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) G[i][6 + j] = pose->T[i * 4 + j];
  double S[9][9];
  memset(S, 0, sizeof(S));
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) S[i][j] = pose->cov[i * 6 + j] * 10000;
  for (int i = 0; i < 3; i++) S[6 + i][6 + i] = 0.1;
  double GS[3][9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 9; j++) {
      double s = 0;
      for (int k = 0; k < 9; k++) s += G[i][k] * S[k][j];
      GS[i][j] = s;
    }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 9; k++) s += GS[i][k] * G[j][k];
      cov_point[i * 3 + j] = s;
    }
  return MALIO_OK;
}
}
