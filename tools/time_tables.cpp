// Developer aid (pure host code, no GPU): what the a15 / f-3 work that stays on the caller's thread costs per scan -
// the pose_unc table chains of laserMapping.cpp:1028-1048 (Mapping::pose_uncertainty_tables: two compounds and one
// inverse compound per entry) and the three IMU propagation tracks of IMU_Processing.hpp:275-389 (malio_predict).
//   g++ -O2 -std=c++17 -I include -I ma-lio_amd/host tools/time_tables.cpp -L ma-lio_amd -lmalio_hip -Wl,-rpath,$PWD/ma-lio_amd -o /tmp/time_tables
#include <chrono>
#include <cstdio>
#include "malio_mapping.hpp"
int main() {
  for (int L : {1, 2, 3}) {
    for (int entries : {10, 20, 40}) {
      std::vector<malio::Pose> extrinsic(L), tc(L > 1 ? L - 1 : 0);
      std::vector<std::vector<malio::Pose>> unc(L);
      auto mk = [](int s) {
        malio::Pose p;
        std::memset(&p, 0, sizeof(p));
        p.q[3] = 1.0;
        p.q[0] = 0.01 * s, p.q[1] = -0.02, p.q[2] = 0.005 * s;
        double nq = std::sqrt(p.q[0] * p.q[0] + p.q[1] * p.q[1] + p.q[2] * p.q[2] + 1.0);
        for (int k = 0; k < 4; k++) p.q[k] /= nq;
        p.t[0] = 0.1 * s, p.t[1] = -0.05, p.t[2] = 0.02 * s;
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 4; c++) p.T[r * 4 + c] = (r == c) ? 1.0 : (c == 3 ? p.t[r] : 0.0);
        p.T[15] = 1.0;
        for (int k = 0; k < 6; k++) p.cov[k * 7] = 1e-6 * (1 + s);
        return p;
      };
      for (int l = 0; l < L; l++) {
        extrinsic[l] = mk(l + 1);
        for (int e = 0; e < entries; e++) unc[l].push_back(mk(e + 3));
      }
      for (auto &p : tc) p = mk(7);
      double sink = 0;
      const int reps = 2000;
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; r++) {
        auto tabs = malio::Mapping::pose_uncertainty_tables(extrinsic, unc, tc);
        sink += tabs[L - 1].back().cov[0];
      }
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
      // propagation: kf.predict + predict_cont + back_predict over `entries` IMU samples
      const int n = 17 + 6 * L;
      std::vector<double> P((size_t)n * n, 0.0), Q(144, 0.0);
      for (int k = 0; k < n; k++) P[(size_t)k * n + k] = 1e-3;
      for (int k = 0; k < 12; k++) Q[k * 13] = 1e-4;
      malio_state_t x;
      std::memset(&x, 0, sizeof(x));
      x.rot[3] = 1.0, x.grav[2] = -9.81;
      for (int l = 0; l < MALIO_MAX_LIDAR; l++) x.offset_R[l][3] = 1.0;
      const double acc[3] = {0.1, -0.2, 9.7}, gyro[3] = {0.01, 0.02, -0.03};
      auto t1 = std::chrono::steady_clock::now();
      for (int r = 0; r < 200; r++)
        for (int k = 0; k < 3 * entries; k++) malio_predict(L, &x, P.data(), 0.0025, Q.data(), acc, gyro);
      double usp = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / 200;
      printf("L=%d entries/LiDAR=%2d: pose_unc tables %7.1f us per scan | 3 x %2d propagation steps %7.1f us per scan (%.2f us per step)  [%g]\n",
             L, entries, us, entries, usp, usp / (3 * entries), sink + x.pos[0]);
    }
  }
  return 0;
}
