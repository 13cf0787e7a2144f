// TEST INFRASTRUCTURE - CPU oracle, voxel down-sampling. Never linked into or called by the product path
// (only tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
//
// The reference calls pcl::VoxelGrid<pcl::PointXYZINormal> (src/laserMapping.cpp:93,860,968-971). PCL is a third-
// party dependency that is NOT under /root/reference (MA_LIO/CMakeLists.txt:56 find_package(PCL 1.8 REQUIRED),
// version unpinned; Ubuntu 20.04 ships 1.10) and is not installed here, so its published algorithm is restated:
//   pcl/filters/impl/voxel_grid.hpp  VoxelGrid<PointT>::applyFilter   (bounds, idx, sort, one output per idx)
//   pcl/common/impl/accumulators.hpp AccumulatorXYZ / Normal / Intensity / Curvature (CentroidPoint)
// PARITY UNPINNED: no PCL build, golden vector or fixture exists to check this against. Cross-checked in
// tests/test_voxel.py against an independent NumPy group-by.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {
struct P12 {
  float v[12];
};
struct IdxPt {
  unsigned int idx;
  unsigned int cloud_point_index;
};
}  // namespace

extern "C" int orc_voxel_downsample(const float *p12, int n, float leaf, int normalize_normal, float *out12, int cap) {
  const P12 *in = (const P12 *)p12;
  // getMinMax3D over finite points
  float min_p[3] = {INFINITY, INFINITY, INFINITY}, max_p[3] = {-INFINITY, -INFINITY, -INFINITY};
  bool any = false;
  for (int i = 0; i < n; i++) {
    if (!std::isfinite(in[i].v[0]) || !std::isfinite(in[i].v[1]) || !std::isfinite(in[i].v[2])) continue;
    any = true;
    for (int a = 0; a < 3; a++) min_p[a] = std::min(min_p[a], in[i].v[a]), max_p[a] = std::max(max_p[a], in[i].v[a]);
  }
  if (!any) return 0;
  const float inverse_leaf_size = 1.0f / leaf;
  // PCL: dx*dy*dz in int64 against INT_MAX. The product wraps (UB) for absurd extents; the test is applied after every
  // factor instead - identical whenever PCL's own arithmetic is defined.
  bool too_small = false;
  std::int64_t cells = 1;
  for (int a = 0; a < 3; a++) {
    const float ext = (max_p[a] - min_p[a]) * inverse_leaf_size;
    if (!(ext < 4.0e9f)) too_small = true;
    if (!too_small) {
      cells *= static_cast<std::int64_t>(ext) + 1;
      if (cells > static_cast<std::int64_t>(INT32_MAX)) too_small = true;
    }
  }
  if (too_small) {  // "Leaf size is too small for the input dataset": output = input
    for (int i = 0; i < n && i < cap; i++) std::memcpy(out12 + (size_t)i * 12, in[i].v, 48);
    return n;
  }
  int min_b[3], max_b[3], div_b[3], divb_mul[3];
  for (int a = 0; a < 3; a++) {
    min_b[a] = static_cast<int>(std::floor(min_p[a] * inverse_leaf_size));
    max_b[a] = static_cast<int>(std::floor(max_p[a] * inverse_leaf_size));
    div_b[a] = max_b[a] - min_b[a] + 1;
  }
  divb_mul[0] = 1, divb_mul[1] = div_b[0], divb_mul[2] = div_b[0] * div_b[1];
  std::vector<IdxPt> index_vector;
  index_vector.reserve(n);
  for (int i = 0; i < n; i++) {
    if (!std::isfinite(in[i].v[0]) || !std::isfinite(in[i].v[1]) || !std::isfinite(in[i].v[2])) continue;
    int ijk0 = static_cast<int>(std::floor(in[i].v[0] * inverse_leaf_size) - static_cast<float>(min_b[0]));
    int ijk1 = static_cast<int>(std::floor(in[i].v[1] * inverse_leaf_size) - static_cast<float>(min_b[1]));
    int ijk2 = static_cast<int>(std::floor(in[i].v[2] * inverse_leaf_size) - static_cast<float>(min_b[2]));
    int idx = ijk0 * divb_mul[0] + ijk1 * divb_mul[1] + ijk2 * divb_mul[2];
    index_vector.push_back({static_cast<unsigned int>(idx), (unsigned int)i});
  }
  // PCL sorts with an unstable integer sort; the order inside one voxel is unspecified there. Stable here.
  std::stable_sort(index_vector.begin(), index_vector.end(), [](const IdxPt &a, const IdxPt &b) { return a.idx < b.idx; });
  int total = 0;
  size_t index = 0;
  while (index < index_vector.size()) {
    size_t i = index + 1;
    while (i < index_vector.size() && index_vector[i].idx == index_vector[index].idx) ++i;
    // CentroidPoint: add() every point of the leaf, then get()
    float xyz[3] = {0, 0, 0}, normal[4] = {0, 0, 0, 0}, intensity = 0, curvature = 0;
    for (size_t li = index; li < i; li++) {
      const float *p = in[index_vector[li].cloud_point_index].v;
      xyz[0] += p[0], xyz[1] += p[1], xyz[2] += p[2];
      normal[0] += p[4], normal[1] += p[5], normal[2] += p[6], normal[3] += p[7];
      intensity += p[8], curvature += p[9];
    }
    const float fn = static_cast<float>(i - index);
    if (total < cap) {
      float *q = out12 + (size_t)total * 12;
      q[0] = xyz[0] / fn, q[1] = xyz[1] / fn, q[2] = xyz[2] / fn, q[3] = 1.0f;
      if (normalize_normal) {
        float z = ((normal[0] * normal[0] + normal[1] * normal[1]) + normal[2] * normal[2]) + normal[3] * normal[3];
        float len = std::sqrt(z);
        if (len > 0) normal[0] /= len, normal[1] /= len, normal[2] /= len, normal[3] /= len;
        q[4] = normal[0], q[5] = normal[1], q[6] = normal[2], q[7] = normal[3];
      } else {
        q[4] = normal[0] / fn, q[5] = normal[1] / fn, q[6] = normal[2] / fn, q[7] = normal[3] / fn;
      }
      q[8] = intensity / fn, q[9] = curvature / fn, q[10] = 0.f, q[11] = 0.f;
    }
    total++;
    index = i;
  }
  return total;
}
