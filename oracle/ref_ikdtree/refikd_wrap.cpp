// TEST INFRASTRUCTURE (oracle/_ref). C wrapper around the REFERENCE's own ikd-Tree, compiled
// from /root/reference/MA_LIO/include/ikd-Tree/ikd_Tree.cpp where it lies (never copied).
// Exposes the call sites SURVEY.md §8(b)-2 lists: Build (laserMapping.cpp:1007),
// Nearest_Search (:586, driven by the same `#pragma omp parallel for` as :559-563),
// Add_Points (:443-444), Delete_Point_Boxes (:223), size (:824), flatten (:1019).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
#include <ikd_Tree.h>
#include <omp.h>
#include <cstring>

using Tree = KD_TREE<pcl::PointXYZINormal>;
using PV = Tree::PointVector;

static PV to_pv(const float *p12, int n) {
  PV v(n);
  if (n) std::memcpy((void *)v.data(), p12, sizeof(pcl::PointXYZINormal) * (size_t)n);
  return v;
}

extern "C" {

void *refikd_create(float downsample) {
  Tree *t = new Tree();  // heap: MANUAL_Q embeds a 1M-entry array (ikd_Tree.h:18,207)
  t->set_downsample_param(downsample);
  return t;
}
void refikd_destroy(void *h) { delete (Tree *)h; }

void refikd_build(void *h, const float *p12, int n) { ((Tree *)h)->Build(to_pv(p12, n)); }

int refikd_size(void *h) { return ((Tree *)h)->size(); }
int refikd_validnum(void *h) { return ((Tree *)h)->validnum(); }

// Batched Nearest_Search; results padded with zeros when fewer than k are found.
int refikd_knn(void *h, const float *q12, int nq, int k, float *out12, float *out_d2, int *out_cnt,
               int nthreads) {
  Tree *t = (Tree *)h;
  if (nthreads < 1) nthreads = 1;
  omp_set_num_threads(nthreads);
#pragma omp parallel for
  for (int i = 0; i < nq; i++) {
    pcl::PointXYZINormal q;
    std::memcpy((void *)&q, q12 + (size_t)i * 12, 48);
    PV near;
    std::vector<float> d2(k);
    t->Nearest_Search(q, k, near, d2);
    int c = (int)near.size();
    out_cnt[i] = c;
    for (int j = 0; j < k; j++) {
      if (j < c) {
        std::memcpy(out12 + ((size_t)i * k + j) * 12, (void *)&near[j], 48);
        out_d2[(size_t)i * k + j] = d2[j];
      } else {
        std::memset(out12 + ((size_t)i * k + j) * 12, 0, 48);
        out_d2[(size_t)i * k + j] = INFINITY;
      }
    }
  }
  return 0;
}

int refikd_add(void *h, const float *p12, int n, int downsample_on) {
  PV v = to_pv(p12, n);
  return ((Tree *)h)->Add_Points(v, downsample_on != 0);
}

int refikd_delete_boxes(void *h, const float *boxes6, int nb) {
  std::vector<BoxPointType> b(nb);
  for (int i = 0; i < nb; i++)
    for (int a = 0; a < 3; a++) {
      b[i].vertex_min[a] = boxes6[i * 6 + a];
      b[i].vertex_max[a] = boxes6[i * 6 + 3 + a];
    }
  return ((Tree *)h)->Delete_Point_Boxes(b);
}

// Valid (non-deleted) points of the tree, in flatten order. Returns the count (may exceed cap).
int refikd_flatten(void *h, float *out12, int cap) {
  Tree *t = (Tree *)h;
  PV().swap(t->PCL_Storage);
  t->flatten(t->Root_Node, t->PCL_Storage, NOT_RECORD);
  int n = (int)t->PCL_Storage.size();
  for (int i = 0; i < n && i < cap; i++) std::memcpy(out12 + (size_t)i * 12, (void *)&t->PCL_Storage[i], 48);
  return n;
}
}
