// TEST INFRASTRUCTURE - CPU oracle, map maintenance. Never linked into or called by the product path
// (only tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
//
// Restates the two ikd-Tree mutators the mapping loop uses, as operations on a flat list of valid points:
//   VoxMap::add           <- KD_TREE::Add_Points            include/ikd-Tree/ikd_Tree.cpp:478-584
//   VoxMap::delete_boxes  <- KD_TREE::Delete_Point_Boxes    include/ikd-Tree/ikd_Tree.cpp:643-669
// The k-d tree itself (balance, lazy deletion, rebuild thread) is an implementation detail of the reference
// and is not restated: the observable result is the SET of valid points, which tests/ compare with the
// reference tree compiled from source (oracle/_ref) on the same call sequence. Within one voxel the stored
// points are visited in insertion order (the tree visits them in traversal order; the outcome only differs
// on exact float ties of the keeper criterion).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace {

struct P12 {
  float v[12];  // pcl::PointXYZINormal layout: x y z _ nx ny nz _ intensity curvature _ _
  float x() const { return v[0]; }
  float y() const { return v[1]; }
  float z() const { return v[2]; }
  float normal_y() const { return v[5]; }
};

struct Box {
  float vmin[3], vmax[3];
};

// ikd_Tree.cpp:1694-1699
float calc_dist(const P12 &a, const P12 &b) {
  float dist = (a.x() - b.x()) * (a.x() - b.x()) + (a.y() - b.y()) * (a.y() - b.y()) + (a.z() - b.z()) * (a.z() - b.z());
  return dist;
}
// ikd_Tree.cpp:1688-1691 (EPSS = 1e-6, ikd_Tree.h:13)
bool same_point(const P12 &a, const P12 &b) {
  return std::fabs(a.x() - b.x()) < 1e-6 && std::fabs(a.y() - b.y()) < 1e-6 && std::fabs(a.z() - b.z()) < 1e-6;
}
// leaf test of Search_by_range / Delete_by_range (ikd_Tree.cpp:1263-1274, :807)
bool in_box(const Box &b, const P12 &p) {
  return b.vmin[0] <= p.x() && b.vmax[0] > p.x() && b.vmin[1] <= p.y() && b.vmax[1] > p.y() && b.vmin[2] <= p.z() &&
         b.vmax[2] > p.z();
}

struct VoxMap {
  float ds;
  std::vector<P12> pts;
  std::vector<char> dead;
  size_t ndead = 0;
  // voxel index (floor(x/ds) per axis) -> indices into pts; a pure accelerator: the box test in add() decides
  std::unordered_map<uint64_t, std::vector<int>> vox;

  static uint64_t key(long ix, long iy, long iz) {
    const long B = 1L << 20;
    return ((uint64_t)(ix + B) & 0x1FFFFF) | (((uint64_t)(iy + B) & 0x1FFFFF) << 21) | (((uint64_t)(iz + B) & 0x1FFFFF) << 42);
  }
  uint64_t key_of(const P12 &p) const {
    return key((long)std::floor(p.x() / ds), (long)std::floor(p.y() / ds), (long)std::floor(p.z() / ds));
  }
  void push(const P12 &p) {
    pts.push_back(p);
    dead.push_back(0);
    if (ds > 0) vox[key_of(p)].push_back((int)pts.size() - 1);
  }
  void kill(int i) {
    if (!dead[i]) dead[i] = 1, ndead++;
  }
  int size() const { return (int)(pts.size() - ndead); }

  int add(const P12 *in, int n, bool downsample_on) {
    int tmp_counter = 0;
    const bool downsample_switch = downsample_on && ds > 0;
    for (int i = 0; i < n; i++) {
      const P12 &pt = in[i];
      if (!downsample_switch) {
        push(pt);
        continue;
      }
      Box box;
      P12 mid{};
      const float c[3] = {pt.x(), pt.y(), pt.z()};
      for (int a = 0; a < 3; a++) {  // :494-502, float/double mix as written there
        box.vmin[a] = std::floor(c[a] / ds) * ds;
        box.vmax[a] = box.vmin[a] + ds;
        mid.v[a] = box.vmin[a] + (box.vmax[a] - box.vmin[a]) / 2.0;
      }
      std::vector<int> storage;  // Downsample_Storage
      // the box test decides; the 26 surrounding index voxels are probed too because for a lattice size that is
      // not a power of two, a point within one ulp of a voxel face can sit in the box of the neighbouring index
      const long ix = (long)std::floor(pt.x() / ds), iy = (long)std::floor(pt.y() / ds), iz = (long)std::floor(pt.z() / ds);
      for (long dz = -1; dz <= 1; dz++)
        for (long dy = -1; dy <= 1; dy++)
          for (long dx = -1; dx <= 1; dx++) {
            auto it = vox.find(key(ix + dx, iy + dy, iz + dz));
            if (it == vox.end()) continue;
            for (int j : it->second)
              if (!dead[j] && in_box(box, pts[j])) storage.push_back(j);
          }
      float min_dist = calc_dist(pt, mid), tmp_dist;
      double min_cov = pt.normal_y();
      P12 result = pt;
      for (int j : storage) {  // :507-527
        tmp_dist = calc_dist(pts[j], mid);
        if (tmp_dist < ds / 8 && min_dist < ds / 8) {
          if (pts[j].normal_y() < min_cov) {
            min_dist = tmp_dist;
            min_cov = pts[j].normal_y();
            result = pts[j];
          }
        } else if (tmp_dist < min_dist) {
          min_dist = tmp_dist;
          min_cov = pts[j].normal_y();
          result = pts[j];
        }
      }
      if (storage.size() > 1 || same_point(pt, result)) {  // :531-537
        for (int j : storage) kill(j);
        push(result);
        tmp_counter++;
      }
    }
    return tmp_counter;
  }

  int delete_boxes(const Box *b, int nb) {
    int cnt = 0;
    for (int k = 0; k < nb; k++)
      for (size_t i = 0; i < pts.size(); i++)
        if (!dead[i] && in_box(b[k], pts[i])) kill((int)i), cnt++;
    return cnt;
  }
};

}  // namespace

extern "C" {
void *orc_vmap_create(float downsample) {
  VoxMap *m = new VoxMap();
  m->ds = downsample;
  return m;
}
void orc_vmap_destroy(void *h) { delete (VoxMap *)h; }
void orc_vmap_build(void *h, const float *p12, int n) {  // Build(): replaces the content (ikd_Tree.cpp:369-397)
  VoxMap *m = (VoxMap *)h;
  m->pts.clear(), m->dead.clear(), m->vox.clear(), m->ndead = 0;
  for (int i = 0; i < n; i++) {
    P12 p;
    std::memcpy(p.v, p12 + (size_t)i * 12, 48);
    m->push(p);
  }
}
int orc_vmap_size(void *h) { return ((VoxMap *)h)->size(); }
int orc_vmap_add(void *h, const float *p12, int n, int downsample_on) {
  std::vector<P12> v(n);
  if (n) std::memcpy((void *)v.data(), p12, 48 * (size_t)n);
  return ((VoxMap *)h)->add(v.data(), n, downsample_on != 0);
}
int orc_vmap_delete_boxes(void *h, const float *boxes6, int nb) {
  std::vector<Box> b(nb);
  for (int i = 0; i < nb; i++)
    for (int a = 0; a < 3; a++) b[i].vmin[a] = boxes6[i * 6 + a], b[i].vmax[a] = boxes6[i * 6 + 3 + a];
  return ((VoxMap *)h)->delete_boxes(b.data(), nb);
}
int orc_vmap_flatten(void *h, float *out12, int cap) {
  VoxMap *m = (VoxMap *)h;
  int k = 0;
  for (size_t i = 0; i < m->pts.size(); i++)
    if (!m->dead[i]) {
      if (k < cap) std::memcpy(out12 + (size_t)k * 12, m->pts[i].v, 48);
      k++;
    }
  return k;
}
}
