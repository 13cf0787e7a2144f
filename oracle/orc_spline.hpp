// TEST INFRASTRUCTURE — CPU oracle (see orc_math.hpp header).
#pragma once
#include "orc_core.hpp"
namespace orc {
void undistort_lidar(std::vector<Pt> &pts, double lidar_beg_time, double lidar_end_time, const Spline &spline,
                     const std::vector<double> &imu_cov_t, const std::vector<std::array<double, 36>> &imu_cov_c,
                     const Pose &extrinsic, const Pose &lt_lidar_frame, std::vector<Pose> &uncertainty);
}
