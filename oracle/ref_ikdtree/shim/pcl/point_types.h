// TEST INFRASTRUCTURE (oracle). Minimal stand-in for <pcl/point_types.h> so that the
// reference's ikd_Tree.{h,cpp} (MA_LIO/include/ikd-Tree/) compiles unmodified, in place,
// without PCL or Eigen on disk. ikd_Tree.h:11,22,56 needs only the field names of
// pcl::PointXYZINormal (48-byte layout) and Eigen::aligned_allocator.
#pragma once
#include <memory>
#include <vector>
namespace pcl {
struct alignas(16) PointXYZINormal {
  float x = 0.f, y = 0.f, z = 0.f, _pad0 = 1.f;
  float normal_x = 0.f, normal_y = 0.f, normal_z = 0.f, _pad1 = 0.f;
  float intensity = 0.f, curvature = 0.f, _pad2 = 0.f, _pad3 = 0.f;
};
static_assert(sizeof(PointXYZINormal) == 48, "PCL layout");
}  // namespace pcl
namespace Eigen {
template <class T>
using aligned_allocator = std::allocator<T>;
}
