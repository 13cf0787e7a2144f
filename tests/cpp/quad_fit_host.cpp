// TEST INFRASTRUCTURE: the host instantiation of ma-lio_amd/csrc/quad_fit.hpp (four lanes emulated in lockstep) against the
// oracle's esti_plane restatement (oracle/orc_capi.cpp: orc_esti_plane, handed in as a function pointer by
// tests/test_quad_fit.py). Built by that test with g++ -O2 -ffp-contract=off.
#include <cstring>
#include "../../ma-lio_amd/csrc/quad_fit.hpp"

using namespace malio::quad;
typedef int (*orc_esti_plane_fn)(const float *near12, float threshold, double cov_threshold, float *pabcd, double *plane_cov);

extern "C" {
// pts: [n][5][3]; returns the number of cases whose plane (4 floats, bit for bit) or inlier verdict differs; first_bad: its index
long quad_fit_compare(const float *pts, long n, float threshold, orc_esti_plane_fn orc, long *first_bad, float *out_quad4,
                      float *out_orc4, int *out_ok2, const float *wny, double cov_threshold) {
  long bad = 0;
  *first_bad = -1;
  for (long i = 0; i < n; i++) {
    const float *p = pts + i * 15;
    QF row0[3], mine[3], pabcd[4];
    for (int j = 0; j < 3; j++) {
      row0[j] = QF(p[j]);
      for (int s = 0; s < 4; s++) mine[j].v[s] = p[3 * (s + 1) + j];
    }
    const QB okq = esti_plane_quad<QF>(row0, mine, QF(threshold), pabcd);
    float near12[5 * 12];
    std::memset(near12, 0, sizeof(near12));
    for (int k = 0; k < 5; k++) near12[12 * k] = p[3 * k], near12[12 * k + 1] = p[3 * k + 1], near12[12 * k + 2] = p[3 * k + 2];
    float po[4];
    double pc = 0;
    // plane_cov: the neighbours' normal_y ride in as a sixth array (pts stays xyz): wny[n][5], or null
    QD ucq(0.0);
    if (wny) {
      for (int k = 0; k < 5; k++) near12[12 * k + 5] = wny[i * 5 + k];
      QF Wm;
      for (int s = 0; s < 4; s++) Wm.v[s] = wny[i * 5 + s + 1];
      ucq = unit_cov_quad<QF, QD>(QD(cov_threshold), QF(wny[i * 5]), Wm);
    }
    const int oko = orc(near12, threshold, cov_threshold, po, &pc);
    bool same = true;
    if (wny)
      for (int s = 0; s < 4; s++)
        if (std::memcmp(&ucq.v[s], &pc, 8) != 0 && !(ucq.v[s] != ucq.v[s] && pc != pc)) same = false;
    for (int s = 0; s < 4; s++) {  // every lane must hold the result
      for (int d = 0; d < 4; d++) {
        const float a = pabcd[d].v[s];
        if (std::memcmp(&a, &po[d], 4) != 0 && !(a != a && po[d] != po[d])) same = false;  // (NaN == NaN for this purpose)
      }
      if ((okq.v[s] ? 1 : 0) != (oko ? 1 : 0) && po[0] == po[0]) same = false;  // (a NaN plane rejects either way: |NaN| > t is false both sides)
    }
    if (!same) {
      if (bad == 0) {
        *first_bad = i;
        for (int d = 0; d < 4; d++) out_quad4[d] = pabcd[d].v[0], out_orc4[d] = po[d];
        out_ok2[0] = okq.v[0], out_ok2[1] = oko;
      }
      bad++;
    }
  }
  return bad;
}
}
