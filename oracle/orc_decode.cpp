// TEST INFRASTRUCTURE - CPU oracle, sensor decode. Never linked into or called by the product
// path (only tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
//
// Sequential restatement of what the reference does between a City-dataset .bin file and `pl_surf`:
//   file_player/src/ROSThread.cpp:776-796,817-833 (Livox records), :947-957 (Ouster records)
//   MA_LIO/src/preprocess.cpp:59-107 (Preprocess::avia_handler), :109-149 (Preprocess::oust64_handler)
// The ROS message containers (livox_ros_driver::CustomMsg, sensor_msgs::PointCloud2 / pcl::fromROSMsg) are plain
// field carriers on this path and are not restated. Parity unpinned against real recordings (none ship with the
// reference); the GPU decoders are compared with this restatement bit for bit.
// the && / || expression of preprocess.cpp:96 is kept exactly as the reference writes it (no added parentheses)
#pragma GCC diagnostic ignored "-Wparentheses"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {
struct CustomPoint {  // livox_ros_driver/CustomPoint.msg
  float x = 0, y = 0, z = 0;
  uint8_t reflectivity = 0, tag = 0, line = 0;
  uint32_t offset_time = 0;
};
struct P12 {
  float x = 0, y = 0, z = 0, p0 = 1.f, normal_x = 0, normal_y = 0, normal_z = 0, p1 = 0, intensity = 0, curvature = 0, p2 = 0,
        p3 = 0;
};
}  // namespace

extern "C" int orc_decode_livox(const unsigned char *rec, int n_rec, int N_SCANS, int point_filter_num, double blind,
                                int eof_point, float *out12, int cap, double *maximum_time_out) {
  // ROSThread.cpp:780-792: one CustomPoint per 19-byte record; `while(!file.eof())` runs once more after the last
  // record, pushing a default-constructed point
  std::vector<CustomPoint> points;
  for (int i = 0; i < n_rec; i++) {
    CustomPoint p;
    const unsigned char *b = rec + (size_t)i * 19;
    std::memcpy(&p.x, b, 4), std::memcpy(&p.y, b + 4, 4), std::memcpy(&p.z, b + 8, 4);
    p.reflectivity = b[12], p.tag = b[13], p.line = b[14];
    std::memcpy(&p.offset_time, b + 15, 4);
    points.push_back(p);
  }
  if (eof_point) points.push_back(CustomPoint());
  // preprocess.cpp:59-107
  const int plsize = (int)points.size();
  std::vector<P12> pl_full(plsize), pl_surf;
  unsigned valid_num = 0;
  double maximum_time = -9999;
  for (unsigned i = 1; i < (unsigned)plsize; i++) {
    if ((points[i].line < N_SCANS) && ((points[i].tag & 0x30) == 0x10 || (points[i].tag & 0x30) == 0x00)) {
      valid_num++;
      if (valid_num % point_filter_num == 0) {
        pl_full[i].x = points[i].x;
        pl_full[i].y = points[i].y;
        pl_full[i].z = points[i].z;
        pl_full[i].intensity = points[i].reflectivity;
        pl_full[i].curvature = points[i].offset_time / float(1000000);
        if (pl_full[i].curvature > 100) continue;
        if (maximum_time < pl_full[i].curvature) maximum_time = pl_full[i].curvature;
        if ((std::abs(pl_full[i].x - pl_full[i - 1].x) > 1e-7) || (std::abs(pl_full[i].y - pl_full[i - 1].y) > 1e-7) ||
            (std::abs(pl_full[i].z - pl_full[i - 1].z) > 1e-7) &&
                (pl_full[i].x * pl_full[i].x + pl_full[i].y * pl_full[i].y + pl_full[i].z * pl_full[i].z > (blind * blind))) {
          pl_surf.push_back(pl_full[i]);
        }
      }
    }
  }
  for (size_t k = 0; k < pl_surf.size() && (int)k < cap; k++) std::memcpy(out12 + k * 12, &pl_surf[k], 48);
  if (maximum_time_out) *maximum_time_out = maximum_time;
  return (int)pl_surf.size();
}

extern "C" int orc_decode_ouster(const unsigned char *rec, int n_rec, int point_filter_num, double blind,
                                 float time_unit_scale, float *out12, int cap, double *maximum_time_out) {
  // ROSThread.cpp:947-957 (the trailing point its eof loop adds is uninitialised there and not reproduced)
  std::vector<P12> pl_surf;
  double maximum_time = -9999;
  for (int i = 0; i < n_rec; i++) {  // preprocess.cpp:120-146
    if (i % point_filter_num != 0) continue;
    const unsigned char *b = rec + (size_t)i * 22;
    float x, y, z, intensity;
    uint32_t t;
    std::memcpy(&x, b, 4), std::memcpy(&y, b + 4, 4), std::memcpy(&z, b + 8, 4), std::memcpy(&intensity, b + 12, 4);
    std::memcpy(&t, b + 18, 4);
    double range = x * x + y * y + z * z;
    if (range < (blind * blind)) continue;
    P12 added_pt;
    added_pt.x = x, added_pt.y = y, added_pt.z = z;
    added_pt.intensity = intensity;
    added_pt.normal_x = 0, added_pt.normal_y = 0, added_pt.normal_z = 0;
    added_pt.curvature = t * time_unit_scale * 1.e-9f;
    if (maximum_time < added_pt.curvature) maximum_time = added_pt.curvature;
    pl_surf.push_back(added_pt);
  }
  for (size_t k = 0; k < pl_surf.size() && (int)k < cap; k++) std::memcpy(out12 + k * 12, &pl_surf[k], 48);
  if (maximum_time_out) *maximum_time_out = maximum_time;
  return (int)pl_surf.size();
}

extern "C" int orc_decode_velodyne(const unsigned char *data, int n_points, int point_step, int off_x, int off_y, int off_z,
                                   int off_intensity, int off_time, int point_filter_num, double blind, float time_unit_scale,
                                   float *out12, int cap, double *maximum_time_io) {
  // Preprocess::velodyne_handler, preprocess.cpp:148-212, on the message's bytes: pcl::fromROSMsg (:155) copies every field
  // of velodyne_ros::Point (preprocess.h:18-34) from the offset the message gives for its name; a field the message lacks
  // stays value-initialised (0).
  std::vector<P12> pl_surf;
  const int plsize = n_points;
  if (plsize == 0) return 0;  // :157-158 (maximum_time keeps its previous value)
  // :161-186 - is_first, yaw_fp, yaw_last, time_last, given_offset_time, yaw_first, yaw_end, layer_first - are written
  // and never read again (grep: given_offset_time has no reader in MA_LIO/src): no observable effect, not restated.
  double maximum_time = -9999;  // :188
  for (int i = 0; i < plsize; i++) {
    const unsigned char *b = data + (size_t)i * (size_t)point_step;
    float x, y, z, intensity = 0.f, time = 0.f;
    std::memcpy(&x, b + off_x, 4), std::memcpy(&y, b + off_y, 4), std::memcpy(&z, b + off_z, 4);
    if (off_intensity >= 0) std::memcpy(&intensity, b + off_intensity, 4);
    if (off_time >= 0) std::memcpy(&time, b + off_time, 4);
    P12 added_pt;
    added_pt.normal_x = 0, added_pt.normal_y = 0, added_pt.normal_z = 0;  // :193-195
    added_pt.x = x, added_pt.y = y, added_pt.z = z;
    added_pt.intensity = intensity;
    added_pt.curvature = time * time_unit_scale;  // :200
    if (i % point_filter_num == 0) {              // :202
      if (added_pt.x * added_pt.x + added_pt.y * added_pt.y + added_pt.z * added_pt.z > (blind * blind)) {  // :204
        if (maximum_time < added_pt.curvature) maximum_time = added_pt.curvature;
        pl_surf.push_back(added_pt);
      }
    }
  }
  for (size_t k = 0; k < pl_surf.size() && (int)k < cap; k++) std::memcpy(out12 + k * 12, &pl_surf[k], 48);
  if (maximum_time_io) *maximum_time_io = maximum_time;
  return (int)pl_surf.size();
}
