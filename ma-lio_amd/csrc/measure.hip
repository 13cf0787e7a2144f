// The per-pass hot path: one h_share_model pass (laserMapping.cpp:552-760) fused with the
// H^T R^-1 H / H^T R^-1 h accumulation of esekfom.hpp:621-635, as gfx950 kernels:
//   k_search         a1-a3 (+a6/a8 trace) of a SEARCH pass: world transform, 5-NN on the two-level neighbour lists,
//                    float plane fit, gates - one kernel, three phases per workgroup (see its header)
//   k_reuse          the same for a REUSE pass (neighbours and plane kept)
//   k_rows_reduce    a5/a7/a10: Jacobian row, FIC weights, per-workgroup 16x16 f64 MFMA outer-product accumulation
//   k_final_reduce   deterministic fixed-order sum of the workgroup partials
//   k_pass           a search (or, in the enqueued-ahead update, reuse) pass WITH its rows in one kernel, speculating on the
//                    previous pass' extrema (rounds 3-5); k_reuse_rows: the reuse pass as one streaming kernel (round 6)
// plus the map_incremental selection (k_far_nearest, k_mapinc_classify) and the batched Nearest_Search (k_nearest).
// Compiled with -ffp-contract=off: the float stages (distances, QR plane fit, gates) and the double
// world transform follow the reference's operation order without FMA contraction, so discrete outcomes
// (neighbour sets, accept flags) are reproducible against the CPU restatement.
#include "malio_internal.hpp"
#include "../host/manifold.hpp"
#include "quad_fit.hpp"

namespace malio {

constexpr int NSUM = 97;  // 78 (12x12 upper) + 12 (rhs) + 6 (c^2 n n^T) + 1 (count)
constexpr u32 INVALID = 0xFFFFFFFFu;
typedef double f64x4 __attribute__((ext_vector_type(4)));

// Developer aid (make PHASE=1): shader-clock stamps of one mid-grid workgroup at phase boundaries of the
// latency-bound kernels; read back with malio_debug_phase(). Compiled out of the shipped library.
#ifdef MALIO_PHASE_CLOCK
__device__ long long g_phase[4][16];
#define PH(kid, k)                                                                     \
  do {                                                                                 \
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) g_phase[kid][k] = wall_clock64(); \
  } while (0)
// grid-wide spread of one kernel: entry and exit time of every workgroup (plain stores: same-address atomics from
// 1.5 k workgroups would serialise for tens of microseconds and distort what they measure)
__device__ long long g_span[14][8192];  // 12: HW_ID, 13: XCC_ID of the workgroup's first wave
//  // entry, exit, end of level-1 search, pending level-2 queries; 4-7: each wave's own end of the level-1 search, 8-11: ... of its directory probe
// the helper wave's stamps: its lane 0 is thread 64
#define PHH(kid, k)                                                                     \
  do {                                                                                  \
    if (threadIdx.x == 64 && blockIdx.x == gridDim.x / 2) g_phase[kid][k] = wall_clock64(); \
  } while (0)
#define PH_ENTER()                                                                    \
  do {                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < 8192) {                                       \
      g_span[0][blockIdx.x] = wall_clock64();                                          \
      g_span[12][blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);              \
      g_span[13][blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);             \
    }                                                                                  \
  } while (0)
#define PH_NOTE(row, val)                                                             \
  do {                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_span[row][blockIdx.x] = (val);       \
  } while (0)
#define PH_WAVE(row0)                                                                  \
  do {                                                                                 \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 8192) g_span[(row0) + (threadIdx.x >> 6)][blockIdx.x] = wall_clock64(); \
  } while (0)
#define PH_EXIT()                                                                     \
  do {                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_span[1][blockIdx.x] = wall_clock64(); \
  } while (0)
#else
#define PH(kid, k)
#define PHH(kid, k)
#define PH_ENTER()
#define PH_NOTE(row, val)
#define PH_WAVE(row0)
#define PH_EXIT()
#endif

// Developer aid: attribution builds (tools/attr_variants.sh; WRONG results, timing and PMC counters only; never shipped):
//   -DATTR_NO_GATHER  phase C takes the five neighbours' coordinates from LDS instead of gathering them from the map array
//   -DATTR_NO_STATE   the per-point state of a search pass (74 B / point) is not stored
//   -DATTR_WALK5      a list walk stops after the first five entries (probe and first line stay)
//   -DATTR_NO_TILES   k_pass does not store its tile
#ifdef ATTR_NO_STATE
#define STATE_ST(stmt) ;
#else
#define STATE_ST(stmt) stmt
#endif

// Developer aid (make POISON=1 -> -DMALIO_POISON; never shipped): the hand-off structures the waves of a workgroup pass each
// other through LDS start every kernel as 0x7FC0DEAD words - a quiet NaN as a float, (NaN, NaN) halves as a double, 2.1 G as a
// map index (beyond every array: the access faults), 0xAD as a neighbour count - instead of what the previous workgroup on the
// CU left there, which is usually a plausible value of the right kind (the probe-cache fault of round 5 was exactly that: a lane
// without a point used a neighbouring workgroup's leftovers as its cached directory probe). The whole GPU suite runs green on this
// build (tools/loop_suite.sh poison).
#ifdef MALIO_POISON
template <class T>
__device__ __forceinline__ void poison_lds(T &obj) {
  u32 *p = reinterpret_cast<u32 *>(&obj);
  for (int k = (int)threadIdx.x; k < (int)(sizeof(T) / 4); k += (int)blockDim.x) p[k] = 0x7FC0DEADu;
}
#define POISON_LDS(obj) poison_lds(obj)
#define POISON_SYNC() __syncthreads()
#else
#define POISON_LDS(obj)
#define POISON_SYNC()
#endif

__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ Q4 qconj(Q4 q) { return {-q.x, -q.y, -q.z, q.w}; }
// Eigen::QuaternionBase::_transformVector order: v + w*uv + qv x uv, uv = 2 (qv x v)
__device__ __forceinline__ D3 qrot(Q4 q, D3 v) {
  D3 qv{q.x, q.y, q.z};
  D3 uv = cross(qv, v);
  uv = uv + uv;
  D3 c2 = cross(qv, uv);
  return {(v.x + q.w * uv.x) + c2.x, (v.y + q.w * uv.y) + c2.y, (v.z + q.w * uv.z) + c2.z};
}
__device__ __forceinline__ D3 mulR(const double *R, D3 v) {
  return {R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z,
          R[6] * v.x + R[7] * v.y + R[8] * v.z};
}
__device__ __forceinline__ D3 mulRt(const double *R, D3 v) {
  return {R[0] * v.x + R[3] * v.y + R[6] * v.z, R[1] * v.x + R[4] * v.y + R[7] * v.z,
          R[2] * v.x + R[5] * v.y + R[8] * v.z};
}

constexpr int MM_SLOTS = 64;  // extrema slots (see mm_publish)
struct Pass1Args {
  int N;
  const float4 *scan;
  // map
  // tables
  const UncEntry *unc;
  int unc_off[MALIO_MAX_LIDAR], unc_len[MALIO_MAX_LIDAR];
  // state
  QuatConst qc;
  float plane_th;
  double cov_threshold;
  int extrinsic_est_en;
  const float4 *map_in;  // [map slots] x y z normal_y (index = map id)
  // per-point outputs (sorted order)
  float4 *world4;  // [N] world point of the search pass
  u64 *mm_cur;     // extrema slots this pass accumulates into (MmSlots)
  u64 *mm_next;    // the other parity: reset here for the next pass
  PartView part;   // spatial shard this handle serves (world <= 1: everything)
  u32 *nbr;  // [5][N] ORIGINAL map indices (INVALID when fewer than 5 inside the radius)
  float4 *plane;
  float *pd2;
  float *world;  // [3][N]
  double *ucov;
  double *trace;
  unsigned char *sel;
  unsigned char *nfound;
  float *ny;        // [N] feats_down_body[i].normal_y as the reference would hold it (committed lazily)
  float4 *cert;     // [N] search-skip certificate: world point of the point's last list walk, radius free of outsiders
  unsigned char *kept;  // [N] this search pass kept the point's cached neighbours (diagnostics)
  uint4 *pcache;    // [N] the point's last level-1 directory probe: cell key (2 words), start and count of the list (search_wg, phase B)
  int skip;         // bit 0: this search pass may keep cached neighbours, bit 1: pcache holds probes of this scan against the
                    // lists as they are, bit 2: this pass writes its probes to pcache (DEV: the control block's search_skip)
  int commit_prev;  // the previous pass was valid: fold its (sel, trace) into ny before overwriting them
  // device loop (DEV = true instantiations): state, parities and commit_prev come from *dl, the slot sets from mm_base
  const DevLoop *dl;
  u64 *mm_base;     // [2 parities][MM_SLOTS][5]
};
// what a DEV kernel reads from the control block instead of from its arguments
struct PassDyn {
  int commit_prev, skip;
  u64 *mm_cur, *mm_next;
};
template <bool DEV>
__device__ __forceinline__ PassDyn pass_dyn(const Pass1Args &a) {
  PassDyn d;
  if (DEV) {
    const int mp = a.dl->mm_parity;
    d.commit_prev = a.dl->commit_prev, d.skip = a.dl->search_skip;
    d.mm_cur = a.mm_base + (size_t)mp * MM_SLOTS * 5, d.mm_next = a.mm_base + (size_t)(mp ^ 1) * MM_SLOTS * 5;
  } else {
    d.commit_prev = a.commit_prev, d.skip = a.skip, d.mm_cur = a.mm_cur, d.mm_next = a.mm_next;
  }
  return d;
}

__device__ __forceinline__ u64 cell_key_d(int ix, int iy, int iz) {
  const long long B = 1ll << 20;
  return ((u64)(ix + B) & 0x1FFFFF) | (((u64)(iy + B) & 0x1FFFFF) << 21) | (((u64)(iz + B) & 0x1FFFFF) << 42);
}
__device__ __forceinline__ u32 hash_key_d(u64 k) {  // must equal hash_key() in map_hash.hip
  u32 lo = (u32)k, hi = (u32)(k >> 32);
  u32 h = lo * 0x9E3779B1u ^ hi * 0x85EBCA77u;
  h ^= h >> 15;
  h *= 0xC2B2AE3Du;
  h ^= h >> 13;
  return h;
}

// Sorted (ascending) 5-slot candidate list with the total order (d2, original map index).
struct Top5 {
  // (d2 bits << 32 | map index): squared distances are >= +0, so their float bits order like the values and ONE
  // 64-bit compare implements the total order (d2, map index). The candidate loop is VALU-issue bound (6 waves per
  // SIMD share one pipe): separate (d2, index) compares and selects needed ~120 instructions per candidate, a
  // compare-exchange chain on packed keys ~45 with two branches, the branch-free form below 36.
  u64 k[5];
  __device__ __forceinline__ float d(int i) const { return __uint_as_float((u32)(k[i] >> 32)); }
  __device__ __forceinline__ u32 og(int i) const { return (u32)k[i]; }
};
__device__ __forceinline__ u64 top5_key(float d2, u32 og) { return ((u64)__float_as_uint(d2) << 32) | (u64)og; }
// the largest key anybody inserts ("no candidate"): low word INVALID; as an f64 bit pattern it is the largest finite
// double, NOT a NaN (see the f64 form of the insertion)
constexpr u64 TOP5_MAXKEY = 0x7FEFFFFFFFFFFFFFull;
// Keys are >= +0 as integers with the sign bit clear and - d2 being a finite float - never carry an all-ones f64
// exponent: read as DOUBLES they are positive finite numbers (denormals included, which the f64 units handle at full
// speed) that order exactly like the integers. A sorted insertion is then five min/max pairs on the f64 pipe: 10
// half-rate instructions instead of 5 64-bit compares and 18 selects.
__device__ __forceinline__ double f64_min_raw(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double f64_max_raw(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// Returns the d2 bits (high word) of the key that fell off the end - the largest of the six: what the search-skip
// certificate (nl_search) folds into its lower bound on the distance of everything OUTSIDE the list.
__device__ __forceinline__ u32 top5_insert(Top5 &t, u64 key) {
  double x = __longlong_as_double((long long)key);
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const double cur = __longlong_as_double((long long)t.k[k]);
    const double lo = f64_min_raw(cur, x);
    x = f64_max_raw(cur, x);
    t.k[k] = (u64)__double_as_longlong(lo);
  }
  return (u32)__double2hiint(x);
}

// One hash probe: (start, count) of cell `key`, (0,0) when the cell is empty.
__device__ __forceinline__ void cell_lookup(const Cell *__restrict__ table, u32 tmask, u64 key, Cell first, u32 slot,
                                            u32 &start, u32 &count) {
  Cell c = first;
  while (true) {
    if (c.key == key) {
      start = c.start, count = c.count;
      return;
    }
    if (c.key == EMPTY_KEY) {
      start = 0, count = 0;
      return;
    }
    slot = (slot + 1) & tmask;
    c = table[slot];
  }
}

// Merge the G lane-local sorted lists into the global top-5 (identical in every lane of the group):
// 5 rounds of a 64-bit (d2 bits | map index) min-reduction over xor-shuffles.
// ev (in: the smallest d2 bits this lane's insertions pushed off its list; out, identical in every lane of the group):
// the smallest d2 bits among ALL candidates of the group that are not in the merged top-5 - the sixth distance.
template <int G, bool CERT>
__device__ __forceinline__ void merge_group(Top5 &t, u32 &ev, u64 refill = TOP5_MAXKEY) {
  Top5 out;
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const u64 key = t.k[0];
    u64 mn = key;
#pragma unroll
    for (int sft = G / 2; sft > 0; sft >>= 1) {
      u64 other = __shfl_xor(mn, sft);
      mn = other < mn ? other : mn;
    }
    out.k[r] = mn;
    if (key == mn && (u32)key != INVALID) {
#pragma unroll
      for (int k = 0; k < 4; k++) t.k[k] = t.k[k + 1];
      t.k[4] = refill;
    }
  }
  if (CERT) {
    ev = min(ev, (u32)(t.k[0] >> 32));  // what is left of this lane's list did not make it either
#pragma unroll
    for (int sft = G / 2; sft > 0; sft >>= 1) ev = min(ev, (u32)__shfl_xor((int)ev, sft));
  }
  t = out;
}

// The same merge for the search's groups of FOUR lanes, as a sorting network on the f64 pipe (round 6; KS_MERGE_NET). The pass
// is bound by VALU issue about as much as by its L1 queue (profiles/round6/r06e_quad_phase_c.txt: +1 900 wave-instructions per
// workgroup cost +4 us), and the five rounds of min-reduction above are ~200 issue slots per lane - as many as a batch's eight
// insertions. Here: a lane and its neighbour (quad_perm [1, 0, 3, 2], then [2, 3, 0, 1]: DPP moves, no LDS crossbar) hold two
// ascending lists A, B; min(A[i], B[4 - i]), i = 0 .. 4, are the five smallest of the ten (the lower half of a bitonic merge),
// and a nine-comparator network puts them in order: 2 x (10 moves + 5 + 18 min / max) ~ 90 slots. Keys are distinct positive
// finite doubles (top5_insert) but for the empty-slot keys, which are equal VALUES: the result is the same list in every lane,
// and the same list as merge_group's.
template <int CTRL>
__device__ __forceinline__ double dpp_quad_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ void merge_pair_net(double k[5]) {
  double b[5];
#pragma unroll
  for (int i = 0; i < 5; i++) b[i] = dpp_quad_f64<CTRL>(k[i]);
  double c[5];
#pragma unroll
  for (int i = 0; i < 5; i++) c[i] = f64_min_raw(k[i], b[4 - i]);
  auto cx = [&](int i, int j) {
    const double lo = f64_min_raw(c[i], c[j]), hi = f64_max_raw(c[i], c[j]);
    c[i] = lo, c[j] = hi;
  };
  cx(0, 1), cx(3, 4), cx(2, 4), cx(2, 3), cx(1, 4), cx(0, 3), cx(0, 2), cx(1, 3), cx(1, 2);
#pragma unroll
  for (int i = 0; i < 5; i++) k[i] = c[i];
}
__device__ __forceinline__ void merge_quad_net(Top5 &t) {
  double k[5];
#pragma unroll
  for (int i = 0; i < 5; i++) k[i] = __longlong_as_double((long long)t.k[i]);
  merge_pair_net<0xB1>(k);  // quad_perm [1, 0, 3, 2]
  merge_pair_net<0x4E>(k);  // quad_perm [2, 3, 0, 1]
#pragma unroll
  for (int i = 0; i < 5; i++) t.k[i] = (u64)__double_as_longlong(k[i]);
}
#ifndef KS_MERGE_NET
#define KS_MERGE_NET 1
#endif

// Eigen ColPivHouseholderQR<Matrix<float,5,3>>::solve(b = -1) restated with static register indexing
// (common_lib.h:174). Same operation order as the CPU restatement; no runtime-indexed arrays.
__device__ __forceinline__ void qr_solve_5x3(float A[5][3], float x[3]) {
  const float eps = 1.1920929e-07f;
  float hC[3];
  int tr[3];
  float nU[3], nD[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; i++) s += A[i][k] * A[i][k];
    nD[k] = sqrtf(s);
    nU[k] = nD[k];
  }
  float maxn = nU[0];
  if (nU[1] > maxn) maxn = nU[1];
  if (nU[2] > maxn) maxn = nU[2];
  float th = maxn * eps;
  float threshold_helper = (th * th) / 5.0f;
  float ndt = sqrtf(eps);
  int nonzero = 3;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int big = k;
    float bigv = nU[k];
#pragma unroll
    for (int j = k + 1; j < 3; j++)
      if (nU[j] > bigv) bigv = nU[j], big = j;
    float big_sq = bigv * bigv;
    if (nonzero == 3 && big_sq < threshold_helper * (float)(5 - k)) nonzero = k;
    tr[k] = big;
#pragma unroll
    for (int j = k + 1; j < 3; j++) {
      if (big == j) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
          float tmp = A[i][k];
          A[i][k] = A[i][j];
          A[i][j] = tmp;
        }
        float tn = nU[k];
        nU[k] = nU[j];
        nU[j] = tn;
        tn = nD[k];
        nD[k] = nD[j];
        nD[j] = tn;
      }
    }
    float tailSq = 0.f;
#pragma unroll
    for (int i = k + 1; i < 5; i++) tailSq += A[i][k] * A[i][k];
    float c0 = A[k][k];
    float tau, beta;
    if (tailSq <= 1.17549435e-38f) {
      tau = 0.f;
      beta = c0;
#pragma unroll
      for (int i = k + 1; i < 5; i++) A[i][k] = 0.f;
    } else {
      beta = sqrtf(c0 * c0 + tailSq);
      if (c0 >= 0.f) beta = -beta;
      float den = c0 - beta;
#pragma unroll
      for (int i = k + 1; i < 5; i++) A[i][k] = A[i][k] / den;
      tau = (beta - c0) / beta;
    }
    hC[k] = tau;
    A[k][k] = beta;
    if (tau != 0.f) {
#pragma unroll
      for (int j = k + 1; j < 3; j++) {
        float tmp = 0.f;
#pragma unroll
        for (int i = k + 1; i < 5; i++) tmp += A[i][k] * A[i][j];
        tmp += A[k][j];
        A[k][j] -= tau * tmp;
#pragma unroll
        for (int i = k + 1; i < 5; i++) A[i][j] -= tau * A[i][k] * tmp;
      }
    }
#pragma unroll
    for (int j = k + 1; j < 3; j++) {
      if (nU[j] != 0.f) {
        float temp = fabsf(A[k][j]) / nU[j];
        temp = (1.f + temp) * (1.f - temp);
        temp = temp < 0.f ? 0.f : temp;
        float r = nU[j] / nD[j];
        float temp2 = temp * (r * r);
        if (temp2 <= ndt) {
          float s = 0.f;
#pragma unroll
          for (int i = k + 1; i < 5; i++) s += A[i][j] * A[i][j];
          nD[j] = sqrtf(s);
          nU[j] = nD[j];
        } else {
          nU[j] *= sqrtf(temp);
        }
      }
    }
  }
  // permutation: perm = identity; swap(perm[k], perm[tr[k]]) for k = 0,1,2
  int p0 = 0, p1 = 1, p2 = 2;
  {
    if (tr[0] == 1) { int t_ = p0; p0 = p1; p1 = t_; }
    if (tr[0] == 2) { int t_ = p0; p0 = p2; p2 = t_; }
    if (tr[1] == 2) { int t_ = p1; p1 = p2; p2 = t_; }
  }
  float c[5] = {-1.f, -1.f, -1.f, -1.f, -1.f};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (k < nonzero) {
      float tau = hC[k];
      if (tau != 0.f) {
        float tmp = 0.f;
#pragma unroll
        for (int i = k + 1; i < 5; i++) tmp += A[i][k] * c[i];
        tmp += c[k];
        c[k] -= tau * tmp;
#pragma unroll
        for (int i = k + 1; i < 5; i++) c[i] -= tau * A[i][k] * tmp;
      }
    }
  }
#pragma unroll
  for (int i = 2; i >= 0; i--) {
    if (i < nonzero) {
      float s = c[i];
#pragma unroll
      for (int j = i + 1; j < 3; j++)
        if (j < nonzero) s -= A[i][j] * c[j];
      c[i] = s / A[i][i];
    }
  }
  float y0 = nonzero > 0 ? c[0] : 0.f, y1 = nonzero > 1 ? c[1] : 0.f, y2 = nonzero > 2 ? c[2] : 0.f;
  x[0] = x[1] = x[2] = 0.f;
  // x[perm[i]] = y_i
  x[0] = (p0 == 0) ? y0 : (p1 == 0) ? y1 : y2;
  x[1] = (p0 == 1) ? y0 : (p1 == 1) ? y1 : y2;
  x[2] = (p0 == 2) ? y0 : (p1 == 2) ? y1 : y2;
  if (nonzero < 3) {  // columns whose pivot position >= nonzero stay 0
    if (p2 == 0) x[0] = 0.f;
    if (p2 == 1) x[1] = 0.f;
    if (p2 == 2) x[2] = 0.f;
    if (nonzero < 2) {
      if (p1 == 0) x[0] = 0.f;
      if (p1 == 1) x[1] = 0.f;
      if (p1 == 2) x[2] = 0.f;
    }
    if (nonzero < 1) x[0] = x[1] = x[2] = 0.f;
  }
}

// trace(Sigma_p) (associate_uct.hpp:153-175 as the caller uses it, laserMapping.cpp:697-699,740-741)
__device__ __forceinline__ double point_trace(const UncEntry &e, float px, float py, float pz) {
  double x = (double)px * 0.05, y = (double)py * 0.05, z = (double)pz * 0.05;
  double a = e.T[0] * x + e.T[1] * y + e.T[2] * z + e.T[3];
  double b = e.T[4] * x + e.T[5] * y + e.T[6] * z + e.T[7];
  double c = e.T[8] * x + e.T[9] * y + e.T[10] * z + e.T[11];
  return e.k0 + (e.lin[0] * a + e.lin[1] * b + e.lin[2] * c) +
         (e.Q[0] * a * a + e.Q[1] * b * b + e.Q[2] * c * c + 2.0 * (e.Q[3] * a * b + e.Q[4] * a * c + e.Q[5] * b * c));
}

// Wave-wide extrema on DPP row operations (quad_perm, row_half_mirror, row_mirror inside the 16-lane rows, row_bcast
// 15 / 31 across them, lane 63 read back): six dependent steps of ~20 cycles. The xor-shuffle form (ds_bpermute: an
// LDS-crossbar round trip per step and per 32-bit half) cost ~1 us for the four reductions at the end of k_search's
// phase C, on every workgroup's critical path.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane63_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
#ifndef KS_NO_DPP
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_f64<0xB1, 0xF>(v));   // quad_perm [1,0,3,2]
  v = fmax(v, dpp_f64<0x4E, 0xF>(v));   // quad_perm [2,3,0,1]
  v = fmax(v, dpp_f64<0x141, 0xF>(v));  // row_half_mirror
  v = fmax(v, dpp_f64<0x140, 0xF>(v));  // row_mirror
  v = fmax(v, dpp_f64<0x142, 0xA>(v));  // row_bcast:15 into rows 1 and 3
  v = fmax(v, dpp_f64<0x143, 0xC>(v));  // row_bcast:31 into rows 2 and 3
  return readlane63_f64(v);
}
__device__ __forceinline__ double wave_min(double v) {
  v = fmin(v, dpp_f64<0xB1, 0xF>(v));
  v = fmin(v, dpp_f64<0x4E, 0xF>(v));
  v = fmin(v, dpp_f64<0x141, 0xF>(v));
  v = fmin(v, dpp_f64<0x140, 0xF>(v));
  v = fmin(v, dpp_f64<0x142, 0xA>(v));
  v = fmin(v, dpp_f64<0x143, 0xC>(v));
  return readlane63_f64(v);
}
#else
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmax(v, __shfl_xor(v, d));
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmin(v, __shfl_xor(v, d));
  return v;
}
#endif

// ---- a1: p' (LiDAR-0 frame) and the world point, double -> float (laserMapping.cpp:569-578) --------
__device__ __forceinline__ void world_point(const QuatConst &qc, const float4 q, int lid, float &wx, float &wy,
                                            float &wz, double &nb) {
  D3 p_body{(double)q.x, (double)q.y, (double)q.z};
  if (lid != 0)
    p_body = qrot(qconj(qc.q0), (qrot(qc.qtc[lid], qrot(qc.ql[lid], p_body) + qc.tl[lid]) + qc.ttc[lid]) - qc.t0);
  D3 pg = qrot(qc.rot, qrot(qc.q0, p_body) + qc.t0) + qc.pos;
  wx = (float)pg.x, wy = (float)pg.y, wz = (float)pg.z;
  nb = sqrt(p_body.x * p_body.x + p_body.y * p_body.y + p_body.z * p_body.z);  // p_body.norm(), :599
}

// esti_plane's plane_cov (common_lib.h:159-173): the five neighbours' normal_y W_j weighted by their distance from cov_threshold
__device__ __forceinline__ double plane_unit_cov(double cov_threshold, const float W[5]) {
  double ucov = 0.0, cov_sum = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) cov_sum += fabs(cov_threshold - (double)W[k]);
  if ((double)W[0] > 0.00001) {
#pragma unroll
    for (int k = 0; k < 5; k++) {
      double wk = (cov_threshold - (double)W[k]) / cov_sum;
      ucov += wk * wk * (double)W[k];
    }
  }
  return ucov;
}

// residual + range gate (laserMapping.cpp:598-601)
// snb = sqrt(|p'|) (the callers form it where it is off the plane fit's dependent chain: phase A of a search pass)
__device__ __forceinline__ bool residual_gate(const float pabcd[4], float wx, float wy, float wz, double snb,
                                              float &pd2) {
  pd2 = pabcd[0] * wx + pabcd[1] * wy + pabcd[2] * wz + pabcd[3];
  float s = (float)(1 - 0.9 * (double)fabsf(pd2) / snb);
  return (double)s > 0.1;
}

// a6/a8: trace(Sigma_p); the index clamp differs for accepted (:694-696) and rejected (:737-739) points
__device__ __forceinline__ double trace_for(const Pass1Args &a, const float4 q, int lid, int tidx, bool selected) {
  // (selects on the kernel arguments: indexing the argument arrays with a lane's slot is a load from the argument segment,
  // and the table entry's load waits for it - one more memory round trip on the control wave's chain)
  int len = a.unc_len[0], off = a.unc_off[0];
#pragma unroll
  for (int l = 1; l < MALIO_MAX_LIDAR; l++)
    if (lid == l) len = a.unc_len[l], off = a.unc_off[l];
  int k = tidx;
  if (selected) {
    if ((unsigned)k >= (unsigned)len) k = len - 2;
  } else {
    if ((unsigned)k >= (unsigned)(len - 1)) k = len - 2;
  }
  if (selected && !a.extrinsic_est_en) return 0.0;  // R(i,0) stays 0, normal_y not rewritten (:681-704)
  return point_trace(a.unc[off + k], q.x, q.y, q.z);
}

// The same under BOTH clamp rules, before the accept flag exists (a helper wave evaluates the trace while the
// control wave fits the plane; the flag then picks one). trS == trace_for(.., true), trR == trace_for(.., false), bit for bit:
// the two rules name different table entries only for the last index of a table.
__device__ __forceinline__ void trace_both(const Pass1Args &a, const float4 q, int lid, int tidx, double &trS, double &trR) {
  int len = a.unc_len[0], off = a.unc_off[0];
#pragma unroll
  for (int l = 1; l < MALIO_MAX_LIDAR; l++)
    if (lid == l) len = a.unc_len[l], off = a.unc_off[l];
  int kS = tidx, kR = tidx;
  if ((unsigned)kS >= (unsigned)len) kS = len - 2;
  if ((unsigned)kR >= (unsigned)(len - 1)) kR = len - 2;
  trR = point_trace(a.unc[off + kR], q.x, q.y, q.z);
  trS = trR;
  if (kS != kR) trS = point_trace(a.unc[off + kS], q.x, q.y, q.z);
  if (!a.extrinsic_est_en) trS = 0.0;  // R(i,0) stays 0 for an accepted point (:681-704)
}

// feats_down_body[i].normal_y bookkeeping (laserMapping.cpp:699,730,741): a pass that reached the end
// rewrites it with trace(Sigma_p), except for accepted points when extrinsic_est_en is off (:681).
// A pass that bailed out with no effective points (:635-639) rewrites nothing, which is only known on the
// host after the pass - so the fold happens at the start of the NEXT pass (or in malio_scan_get).
__device__ __forceinline__ void commit_normal_y(const Pass1Args &a, int commit_prev, int i) {
  if (!commit_prev) return;
  if (a.sel[i] && !a.extrinsic_est_en) return;
  a.ny[i] = (float)a.trace[i];
}

// a4: min/max of unit_cov and R over the accepted points, and their count (laserMapping.cpp:614-632,646-647).
// Extrema and an integer count are order-independent, so they are combined with atomics - spread over MM_SLOTS
// addresses so that no address sees more than a few dozen (same-address atomics serialise at ~12 ns each) - and
// folded by one wave of the consumer. Slot layout: [MM_SLOTS][5] u64 = max_u, min_u, max_R, min_R (doubles under
// an order-preserving encoding), count. Two parities alternate between passes; a stage-1 kernel accumulates into
// one and resets the other, so no separate clearing launch is needed.
__device__ __forceinline__ u64 mm_enc(double x) {
  u64 b = (u64)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double mm_dec(u64 k) {
  u64 b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}
__device__ __forceinline__ void mm_reset_slot(u64 *slots, int sl) {
  u64 *o = slots + (size_t)sl * 5;
  o[0] = mm_enc(-INFINITY), o[1] = mm_enc(INFINITY), o[2] = mm_enc(-INFINITY), o[3] = mm_enc(INFINITY), o[4] = 0;
}
// one lane publishes what its wave / workgroup reduced
__device__ __forceinline__ void mm_publish(u64 *slots, double mxu, double mnu, double mxr, double mnr, u64 count) {
  if (count == 0) return;  // nothing accepted here: the identities would not change anything
  u64 *o = slots + (size_t)(blockIdx.x & (MM_SLOTS - 1)) * 5;
  atomicMax(&o[0], mm_enc(mxu)), atomicMin(&o[1], mm_enc(mnu));
  atomicMax(&o[2], mm_enc(mxr)), atomicMin(&o[3], mm_enc(mnr));
  atomicAdd(&o[4], count);
}
// Called by the first wave (64 lanes) of a consumer: folds the slots with the reference's initial values
// (laserMapping.cpp:615-616,646-647). Lane 0 returns [max_u, -min_u, max_R, -min_R, M] in out5.
__device__ __forceinline__ void mm_fold_wave(const u64 *slots, int extrinsic_est_en, double out5[5]) {
  const int lane = threadIdx.x & 63;
  const u64 *o = slots + (size_t)lane * 5;
  double r0 = mm_dec(o[0]), r1 = mm_dec(o[1]), r2 = mm_dec(o[2]), r3 = mm_dec(o[3]);
  u64 cnt = o[4];
  r0 = wave_max(r0), r1 = wave_min(r1), r2 = wave_max(r2), r3 = wave_min(r3);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
  double m0 = fmax(0.0, r0), m1 = fmin(1000.0, r1), m2 = 0.0, m3 = 9999.0;
  if (extrinsic_est_en) m2 = fmax(m2, r2), m3 = fmin(m3, r3);
  out5[0] = m0, out5[1] = -m1, out5[2] = m2, out5[3] = -m3, out5[4] = (double)cnt;
}

// workgroup-wide version for the thread-per-point kernels (k_reuse)
__device__ __forceinline__ void block_minmax(const Pass1Args &a, const PassDyn &dy, bool selected, double ucov, double tr) {
  __shared__ double sm[BLK / 64][5];
  double mxu = selected ? ucov : -INFINITY, mnu = selected ? ucov : INFINITY;
  bool rsel = selected && a.extrinsic_est_en;
  double mxr = rsel ? tr : -INFINITY, mnr = rsel ? tr : INFINITY;
  mxu = wave_max(mxu), mnu = wave_min(mnu), mxr = wave_max(mxr), mnr = wave_min(mnr);
  unsigned long long bal = __ballot(selected);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm[wave][0] = mxu, sm[wave][1] = mnu, sm[wave][2] = mxr, sm[wave][3] = mnr, sm[wave][4] = (double)__popcll(bal);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double r0 = sm[0][0], r1 = sm[0][1], r2 = sm[0][2], r3 = sm[0][3], r4 = sm[0][4];
#pragma unroll
    for (int w = 1; w < BLK / 64; w++) {
      r0 = fmax(r0, sm[w][0]), r1 = fmin(r1, sm[w][1]), r2 = fmax(r2, sm[w][2]), r3 = fmin(r3, sm[w][3]);
      r4 += sm[w][4];
    }
    mm_publish(dy.mm_cur, r0, r1, r2, r3, (u64)r4);
  }
  if (blockIdx.x == 0 && threadIdx.x < MM_SLOTS) mm_reset_slot(dy.mm_next, threadIdx.x);
}

// a2: ikdtree.Nearest_Search (laserMapping.cpp:586) on neighbour lists. ONE directory probe + one contiguous
// list per query; G lanes stride over the list with 8 independent 16-byte loads in flight each, keep a sorted
// top-5 under the total order (d2, map index) and merge it with 64-bit min-reductions over xor-shuffles.
// The list holds every map point of the 3x3x3 block of cells (edge cf) around the query's cell, so the result is
// exact whenever the 5th distance lies inside the radius that block guarantees at the query's position.
//   level 1 (FINAL = false, cf1 small): queries that cannot be certified are marked NF_PENDING;
//   level 2 (FINAL = true, cf2 >= sqrt 5): guaranteed radius >= cf2 covers the reference's acceptance radius
//   (`pointSearchSqDis[4] > 5` rejects, :587), so whatever it finds inside d2 <= 5 is final.
// Squared distances are computed as ikd_Tree.cpp:1697 without FMA; candidates with d2 > limit2 are dropped.
#ifndef KS_G
#define KS_G 4
#endif
constexpr int NL1_G = KS_G;  // lanes per query on the level-1 lists (~45 candidates, 8 loads in flight per lane)
constexpr unsigned char NF_PENDING = 0xFF;
#ifndef KS_EARLY
#define KS_EARLY 1
#endif
#ifndef KS_PCACHE
#define KS_PCACHE 1
#endif
#ifndef KS_NB
#define KS_NB 8  // 16-byte loads per lane and batch of the level-1 walk (x NL1_G lanes: the entries a sorted list's walk reads before it may stop)
#endif
constexpr unsigned char NF_NOTMINE = 0xFD;   // partitioned handle: the point's tile belongs to another shard
#ifndef KS_L2G
#define KS_L2G 4
#endif
// position of the n-th (0-based) set bit of m; popcount(m) > n
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int n) {
  int pos = 0;
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    const int c = __popcll(m & ((1ull << w) - 1ull));
    if (n >= c) n -= c, m >>= w, pos += w;
  }
  return pos;
}
struct NlView {
  const Cell *table;
  u32 tmask;
  const float4 *pts;
  float cf, inv_cf;
  float reach_cf;  // the radius around the query's CELL the list is complete for: cf, or NL_REACH x cf for a pruned list
  int early;       // walks of an ordered list may end after the first batch (nl_walk, EARLY): set for scans that fill the GPU
};
// lb2 (out, identical in every lane of the group): a lower bound on the SQUARED distance - as this function computes
// distances - from the query to every map point that is not one of the returned neighbours: the smallest of (i) the
// candidates of the list that fell off the top-5 or lie beyond limit2 and (ii) the radius the block guarantees. It is
// what lets a later search pass keep the neighbours without walking the list again (search_wg, phase A').
// CERT = false: the walk of rounds 1-3 (the list starts out holding five (limit + 1 ulp, no index) keys, candidates beyond the
// limit never enter it, nothing is tracked for lb2, which reads 0): every instruction per candidate counts in this loop -
// the bookkeeping of the certificate costs the search pass 0.7 us at BASELINE config 2 (profiles/round4/r04h_*) - so only
// the kernels of a handle with MALIO_OPT_SEARCH_SKIP on carry it.
// The walk of the LONG lists (level 2: 180-900 entries; PIPE) is software-pipelined (KS_PIPE, round 5): BASELINE config 5's
// search pass -1.3 us. Under load a dependent memory round trip of this phase costs ~2 us (all workgroups of the one resident
// generation issue the same burst at the same time); the walk of rounds 1-4 - 8 loads per lane, wait, 8 insertions, again -
// paid every round trip and every batch of insertions one after the other. Here: rounds of THREE 16-byte loads per lane in
// two register sets, the next round's loads issued (and pinned by a scheduling barrier: left alone the scheduler sinks them
// behind the insertions) before this round's candidates are inserted; loads return in order, so the compiler's own
// `s_waitcnt vmcnt(3)` in front of the insertions means "this round has arrived". What keeps it inside the 72-VGPR budget,
// where round 3's attempt spilled: ONE 64-bit pointer per lane with the round's entries at immediate offsets - no clamped
// index, no address per load: a lane whose list has ended keeps its pointer (it loads the same lines again and offers
// MAXKEY), a round's entries past the end of a list read the list's slack, the next list or the guard behind the array
// (build_nlist: NL_GUARD) and offer nothing either -, no SLP vectorisation (Makefile), three loads per round (four: 31
// spilled VGPRs, 43 us). The loop is wave-uniform (a ballot decides). (Loads as inline assembly with hand-placed waits were
// tried first and are a trap: the compiler copies a register set whose loads are still in flight.)
#ifndef KS_PIPE
#define KS_PIPE 1
#endif
// the directory probe alone: (start, count) of the list of the cell that holds (wx, wy, wz); (0, 0): no such cell
__device__ __forceinline__ void nl_probe(const NlView &nl, float wx, float wy, float wz, u32 &start, u32 &count) {
  const float gx = wx * nl.inv_cf, gy = wy * nl.inv_cf, gz = wz * nl.inv_cf;
  const u64 key = cell_key_d((int)floorf(gx), (int)floorf(gy), (int)floorf(gz));
  const u32 slot = hash_key_d(key) & nl.tmask;
  cell_lookup(nl.table, nl.tmask, key, nl.table[slot], slot, start, count);
#ifdef ATTR_WALK5
  count = min(count & NL_COUNT, 5u);
#endif
}
// the walk of a list whose (start, count) are known
// count: as the directory holds it (NL_SORTED in its top bit).
// EARLY (level 1, round 5): a list flagged NL_SORTED is in order of distance from the centre of its cell (map_hash.hip:
// k_nl_sort). Its first batch - 8 G entries: four 128-byte lines - is read as ever; everything behind it lies at least as
// far from the centre as the farthest entry of the batch, r, hence at least r - |query - centre| from the query. If the fifth
// distance so far is below that, no unread entry can enter the five (nor tie with one: the margins below dwarf the float
// rounding of all three distances, 3e-7 relative) and the walk ends: the lines that are never requested are what the search
// pass is short of (outstanding L1 misses per CU). Otherwise (2.5 % of the queries at BASELINE config 2: the ones near a corner
// of their cell) the same lanes read the rest of the list on top of what they have.
template <int G, bool CERT, bool PIPE, bool EARLY = false>
__device__ __forceinline__ bool nl_walk(const NlView &nl, float wx, float wy, float wz, int sub, float limit2, u32 start, u32 count,
                                        Top5 &t, float &lb2) {
  const bool in_order = (count & NL_SORTED) != 0;
  count &= NL_COUNT;
  const bool cut = EARLY && nl.early && in_order && count > (u32)(KS_NB * G);  // (the G lanes of a query agree)
  const u32 cn = cut ? (u32)(KS_NB * G) : count;                   // entries this walk reads
  float r2 = 0.f;  // cut: squared distance from the cell centre of this lane's LAST entry of the batch (0: a tombstone)
  const float sentinel = __uint_as_float(__float_as_uint(limit2) + 1u);  // next float above the limit
#pragma unroll
  for (int k = 0; k < 5; k++) t.k[k] = CERT ? TOP5_MAXKEY : top5_key(sentinel, INVALID);
  u32 ev = (u32)(TOP5_MAXKEY >> 32);
  float gx = wx * nl.inv_cf, gy = wy * nl.inv_cf, gz = wz * nl.inv_cf;
  float kxf = floorf(gx), kyf = floorf(gy), kzf = floorf(gz);
  constexpr int NB = KS_NB;  // loads per lane and batch
  auto batch = [&](u32 j) {  // NB entries of this lane, the first one at j
    float4 m[NB];
#pragma unroll
    for (int u = 0; u < NB; u++) m[u] = nl.pts[(size_t)start + min(j + (u32)(u * G), count - 1)];
    if (EARLY && cut) {  // (a cut list is longer than the batch: entry sub + 7 G exists; a tombstone's x is +inf)
      const float cdx = m[NB - 1].x - (kxf + 0.5f) * nl.cf, cdy = m[NB - 1].y - (kyf + 0.5f) * nl.cf, cdz = m[NB - 1].z - (kzf + 0.5f) * nl.cf;
      const float c2 = cdx * cdx + cdy * cdy + cdz * cdz;  // == map_hash.hip: nl_centre_d2
      r2 = c2 < INFINITY ? c2 : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NB; u++) {
      float ddx = wx - m[u].x, ddy = wy - m[u].y, ddz = wz - m[u].z;
      float d2 = ddx * ddx + ddy * ddy + ddz * ddz;  // calc_dist, ikd_Tree.cpp:1697 (no FMA)
      // A slot past the end of the list (its load was clamped to the last entry) gets the largest key and sorts after
      // everything. CERT: the range test (d2 <= limit2) is applied to the five survivors below, not per candidate; else
      // d2 <= limit2 <=> key < the sentinel keys the list starts with: no range test at all.
      const u64 key = top5_key(d2, __float_as_uint(m[u].w));
      const u32 out = top5_insert(t, j + (u32)(u * G) < count ? key : TOP5_MAXKEY);
      if (CERT) ev = min(ev, out);
    }
  };
  // (the SKIP kernels' walk - CERT: the certificate's bookkeeping rides on every insertion - stays the two-batch loop: with
  // it the pipelined form spills 200 VGPRs; the option is off by default)
  if constexpr (PIPE && !CERT) {
    const float4 *lst = nl.pts + (size_t)start;
    // one candidate slot: entry e of the list as loaded into m (a slot past the end - its load was clamped to the last entry -
    // offers the largest key, which sorts behind everything)
    auto offer = [&](const float4 &m, u32 e) {
      float ddx = wx - m.x, ddy = wy - m.y, ddz = wz - m.z;
      float d2 = ddx * ddx + ddy * ddy + ddz * ddz;  // calc_dist, ikd_Tree.cpp:1697 (no FMA)
      const u64 key = top5_key(d2, __float_as_uint(m.w));
      const u32 out = top5_insert(t, e < count ? key : TOP5_MAXKEY);
      if (CERT) ev = min(ev, out);
    };
#ifndef KS_PIPE_R
#define KS_PIPE_R 3
#endif
    constexpr int R = KS_PIPE_R;  // loads per lane and round
    auto offer_round = [&](const float4 (&m)[R], u32 e) {
#pragma unroll
      for (int u = 0; u < R; u++) {
        offer(m[u], e + (u32)(u * G));
      }
    };
    const float4 *pa = lst + sub;  // this lane's entry of round a (the other entries of the round: immediates)
    u32 j = (u32)sub;              // ... its index in the list
    float4 a[R], b[R];
#pragma unroll
    for (int u = 0; u < R; u++) a[u] = pa[u * G];
    while (true) {
      // (wave-uniform: does any list of this wave reach into the next round?)
      const u32 n = j + (u32)(R * G);
      if (__ballot(n < count) == 0ull) {
        offer_round(a, j);
        break;
      }
      const float4 *pb = pa + (n < count ? R * G : 0);
#pragma unroll
      for (int u = 0; u < R; u++) b[u] = pb[u * G];
      __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the loads behind the insertions they are to overlap with)
      offer_round(a, j);
      j = n + (u32)(R * G);  // (the round after b's)
      if (__ballot(j < count) == 0ull) {
        offer_round(b, n);
        break;
      }
      pa = pb + (j < count ? R * G : 0);
#pragma unroll
      for (int u = 0; u < R; u++) a[u] = pa[u * G];
      __builtin_amdgcn_sched_barrier(0);
      offer_round(b, n);
    }
    if (KS_MERGE_NET && G == 4 && !CERT) merge_quad_net(t);
    else if (G > 1) merge_group<G, CERT>(t, ev, top5_key(sentinel, INVALID));
  } else {
  // The short lists of level 1 (~44 entries: two batches) keep the two-batch walk: pipelined rounds were measured 0.6-1 us
  // SLOWER there (a round's insertions, ~1 us for the SIMD's seven waves, do not cover a ~2 us round trip; rounds of 2 / 3 / 4
  // loads: 30.7 / 30.9 / 43 us - the last one spills - against 29.8 us), and touching the second batch's lines while the
  // first is in flight (an L2 prefetch by 4-byte loads) changed nothing (profiles/round5/r05d_walk_variants.txt).
  // ONE copy of the batch and of the merge in the code (the continuation as a second inlined copy made the kernel 1.3 us
  // slower on lists that are not cut at all - its instruction footprint): a loop of at most two turns, the second one only for
  // the lanes of a cut list whose first batch settled nothing.
  u32 j = (u32)sub, end = cn;
  bool again = EARLY && cut;  // (the G lanes of a query agree on everything that steers this loop)
  while (true) {
    for (; j < end; j += NB * G) batch(j);
    if (KS_MERGE_NET && G == 4 && !CERT) merge_quad_net(t);
    else if (G > 1) merge_group<G, CERT>(t, ev, top5_key(sentinel, INVALID));
    if (!(EARLY && again)) break;
    again = false;
#pragma unroll
    for (int sft = G / 2; sft > 0; sft >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, sft));  // (any entry that was read bounds the unread ones)
    const float qx = wx - (kxf + 0.5f) * nl.cf, qy = wy - (kyf + 0.5f) * nl.cf, qz = wz - (kzf + 0.5f) * nl.cf;
    const float lo = sqrtf(r2) * 0.9999f - sqrtf(qx * qx + qy * qy + qz * qz) * 1.0001f - 1e-6f;
    const float b2 = lo * lo * 0.9999f;  // every unread entry is farther than this (squared)
    if (lo > 0.f && t.og(4) != INVALID && t.d(4) < b2) {
      if (CERT) ev = min(ev, __float_as_uint(b2));  // the unread entries are outsiders as well
      break;
    }
    // not settled (2.5 % of the queries of BASELINE config 2: the ones near a corner of their cell): the rest of the list, on
    // top of what the group has - the merged five stay in the group's first lane, the others start empty again
    if (sub != 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) t.k[k] = CERT ? TOP5_MAXKEY : top5_key(sentinel, INVALID);
    }
    j = (u32)(NB * G + sub), end = count;
  }
  }
  // CERT: survivors beyond the limit are not results: they read (sentinel, INVALID) as an empty slot always has, and count
  // as outsiders for the bound (the list is sorted: they form its tail)
  if (CERT) {
#pragma unroll
    for (int k = 0; k < 5; k++)
      if (!(t.d(k) <= limit2)) {
        ev = min(ev, (u32)(t.k[k] >> 32));
        t.k[k] = top5_key(sentinel, INVALID);
      }
  }
  // radius the block guarantees: one cell edge plus the distance to the nearest face of the own cell, minus a
  // conservative allowance for the float rounding of the cell coordinates
  float fx = gx - kxf, fy = gy - kyf, fz = gz - kzf;
  float margin = 3e-7f * (fabsf(gx) + fabsf(gy) + fabsf(gz) + 3.0f) * nl.cf;
  float fmin_ = fminf(fminf(fminf(fx, 1.f - fx), fminf(fy, 1.f - fy)), fminf(fz, 1.f - fz));
  float g1 = nl.reach_cf + fmaxf(fmin_ * nl.cf - margin, 0.f) - margin;
  const float g2 = g1 * g1 * 0.99999f;
  lb2 = CERT ? __uint_as_float(min(ev, __float_as_uint(g2))) : 0.f;  // (positive floats order like their bits; ev may be the no-candidate mark)
  return (t.og(4) != INVALID) && (t.d(4) <= g2);
}
// (Measured and not kept, profiles/round5/r05l_probe_and_order.txt: the G lanes of a query loading the 2 G directory slots behind
// the home slot in one trip - a wave of 16 queries otherwise waits for its unluckiest linear probe: 93 % of the waves need a
// second dependent slot, 56 % a third - costs k_pass +1.4 us at config 2: twice the probe's requests outweigh the saved trips;
// and the level-1 probe moved into phase A with the queries handed to phase B sorted by the number of batches they need
// (a wave walks as many batches as its longest list: 2.8 where a query alone needs 1.9): -25 % candidate slots, +0.5 us.)
template <int G, bool CERT, bool PIPE, bool EARLY = false>
__device__ __forceinline__ bool nl_search(const NlView &nl, float wx, float wy, float wz, int sub, float limit2,
                                          Top5 &t, float &lb2) {
  u32 start, count;
  nl_probe(nl, wx, wy, wz, start, count);
  return nl_walk<G, CERT, PIPE, EARLY>(nl, wx, wy, wz, sub, limit2, start, count, t, lb2);
}
// lb2 -> the certificate's radius: a lower bound on the TRUE distance of every outsider (computed squared distances are
// within 3e-7 relative of the true ones; sqrtf is correctly rounded)
__device__ __forceinline__ float cert_radius(float lb2) { return sqrtf(lb2) * 0.99999f; }

// a3 + gates of ONE query whose neighbours are known (phase C of the search phases, control wave): plane, accept flag and
// residual go back to the caller, which hands them to the helper wave through LDS - unit_cov, the trace, the per-point state's
// stores and the extrema are the helper's (helper_unit_cov_trace / helper_post). q: the scan point (phase A read it; it comes
// back from LDS, not from HBM). The five neighbours' coordinates are kept in LDS across the fit (S.nbp: the plane fit overwrites
// its copy, the inlier test afterwards reads them from there instead of gathering them a second time). What phase A left in
// LDS - world point, |p'|, scan point - is read where it is used: held in registers across the plane fit it is what the
// register allocator spills.
struct SearchLds;
__device__ __forceinline__ void point_phase(const Pass1Args &a, SearchLds &S, int lane, const u32 og[5], int nf, bool &selected,
                                            float4 &pl_out, float &pd2_out, float4 &q_out);
// REUSE pass of one point (ekfom_data.converge == false, laserMapping.cpp:583-595): neighbours, plane and flag are kept; the
// residual and the range gate are re-evaluated at the new state.
__device__ __forceinline__ void reuse_point(const Pass1Args &a, const QuatConst &qc, int commit_prev, int i, bool &selected,
                                            double &ucov, double &tr, float4 &pl_out, float &pd2_out, float4 &q_out) {
  selected = false, ucov = 0.0, tr = 0.0;
  pl_out = make_float4(0.f, 0.f, 0.f, 0.f), pd2_out = 0.f, q_out = pl_out;
  if (i >= a.N || a.nfound[i] == NF_NOTMINE) return;  // (a partitioned handle keeps serving the points of its last search pass)
  const float4 q = a.scan[i];
  q_out = q;
  const int packed = __float_as_int(q.w);
  const int lid = packed & 0xFF, tidx = packed >> 8;
  float wx, wy, wz;
  double nb;
  world_point(qc, q, lid, wx, wy, wz, nb);
  a.world[i] = wx, a.world[a.N + i] = wy, a.world[2 * a.N + i] = wz;
  commit_normal_y(a, commit_prev, i);
  if (a.sel[i]) {
    const float4 pl = a.plane[i];
    pl_out = pl;
    const float pabcd[4] = {pl.x, pl.y, pl.z, pl.w};
    ucov = a.ucov[i];
    float pd2;
    if (residual_gate(pabcd, wx, wy, wz, sqrt(nb), pd2)) {
      selected = true;
      a.pd2[i] = pd2;
      pd2_out = pd2;
    }
  }
  a.sel[i] = selected ? 1 : 0;
  tr = trace_for(a, q, lid, tidx, selected);
  a.trace[i] = tr;
}
// The same for the control wave of k_pass: loads and arithmetic only. The world point, the flag and the
// residual go back to the caller, which hands them to the helper wave through LDS; the helper stores them with the rest of
// the per-point state (helper_post), forms the trace and folds the previous pass' normal_y.
__device__ __forceinline__ void reuse_point_ctrl(const Pass1Args &a, const QuatConst &qc, int i, bool &selected, float4 &pl_out,
                                                 float &pd2_out, float4 &q_out, float4 &world_out) {
  selected = false;
  pl_out = make_float4(0.f, 0.f, 0.f, 0.f), pd2_out = 0.f, q_out = pl_out, world_out = pl_out;
  if (i >= a.N || a.nfound[i] == NF_NOTMINE) return;
  const float4 q = a.scan[i];
  const unsigned char sel_old = a.sel[i];
  const float4 pl = a.plane[i];  // (always allocated, initialised by the scan's installation: loaded beside the flag, not behind it)
  q_out = q;
  float wx, wy, wz;
  double nb;
  world_point(qc, q, __float_as_int(q.w) & 0xFF, wx, wy, wz, nb);
  world_out = make_float4(wx, wy, wz, 0.f);
  if (sel_old) {
    pl_out = pl;
    const float pabcd[4] = {pl.x, pl.y, pl.z, pl.w};
    float pd2;
    if (residual_gate(pabcd, wx, wy, wz, sqrt(nb), pd2)) {
      selected = true;
      pd2_out = pd2;
    }
  }
}

// a4 over one wave's 64 points: extrema of unit_cov / R and the count of accepted points -> one slot
__device__ __forceinline__ void wave_minmax_publish(const Pass1Args &a, u64 *mm_cur, bool selected, double ucov, double tr) {
  double mxu = selected ? ucov : -INFINITY, mnu = selected ? ucov : INFINITY;
  const bool rsel = selected && a.extrinsic_est_en;
  double mxr = rsel ? tr : -INFINITY, mnr = rsel ? tr : INFINITY;
  mxu = wave_max(mxu), mnu = wave_min(mnu), mxr = wave_max(mxr), mnr = wave_min(mnr);
  const unsigned long long bal = __ballot(selected);
  if ((threadIdx.x & 63) == 0) mm_publish(mm_cur, mxu, mnu, mxr, mnr, (u64)__popcll(bal));
}

// ---- SEARCH pass, one kernel (laserMapping.cpp:563-612 + the rejected-point trace of :725-743) ----------------
// A workgroup owns SQ = 64 consecutive sorted queries and runs three phases:
//   A  wave 0, lane = query: a1 world transform (double, Eigen's operation order) -> LDS, world4, |p'|
//   B  all 4 waves, G = 4 lanes per query: a2 level-1 neighbour-list search (8 x 16-byte loads in flight per lane),
//      result (5 map ids or "not certified") -> LDS; waves 1-3 retire
//   B' all 4 waves: level-2 search for the uncertified queries, one query per wave at a time (64 lanes per query)
//   C  wave 0, lane = query: a3 plane fit, gates, a6/a8 trace, extrema -> slots
// One kernel instead of three saves two kernel boundaries (each costs a few us of drain + cache writeback at
// this size) and lets the latency-bound plane fit of one workgroup overlap with the memory-bound search of the
// others on the same CU.
constexpr int SQ = 64;             // queries per workgroup: one per lane of wave 0 in phases A and C
constexpr int KS_BLK = SQ * NL1_G;  // workgroup size of k_search
#ifndef KS_WPE
#define KS_WPE 7
#endif
// what the control wave (wave 0) of a workgroup knows about its lane's query after the point phase
struct PointOut {
  bool selected = false;
  bool skipped = false;  // the whole workgroup belongs to other shards: nothing was searched
  double ucov = 0.0, tr = 0.0;
  float4 pl = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};  // plane (n, d) and the scan point (body frame, packed slot word): what the row of a5 is built from
  float pd2 = 0.f;
};
struct SearchLds {
  float4 w[SQ];
  u32 og[5][SQ];
  unsigned char nf[SQ];
  unsigned char keep[SQ];  // phase A': the cached neighbours are certified for the new world point - no walk
  int flags;               // what the control wave tells the others after phase A (search_wg)
  float cr[SQ];            // certificate radius of the walk that served the query (cert_radius)
  uint4 pc[SQ];            // the point's cached directory probe (phase A read it with the scan point: one coalesced load)
  float4 q[SQ];            // the scan point as phase A read it (phase C: the row and the trace are built from it)
  float4 (*nbp)[SQ];       // phase C: [5][SQ] - the five neighbours' map points, kept across the plane fit (point_phase); the
                           // kernel's own LDS (k_pass: the storage of its row staging U, which is written after the fit)
  double nb[SQ];  // sqrt(|p'|) of phase A, consumed by the range gate in phase C
  // what the helper wave and the control wave hand each other across their one barrier
  double trS[SQ], trR[SQ];  // helper -> itself: the trace under the accepted / rejected point's clamp rule (trace_both)
  double ucv[SQ];           // helper -> itself: unit_cov of the five neighbours (esti_plane's plane_cov)
  unsigned char selc[SQ];   // control -> helper: the accept flag
  float4 plc[SQ];           // control -> helper: the plane (n, d) - the helper stores the per-point state
  float pd2c[SQ];           // control -> helper: the residual
};
// the plane-independent factors of the a5/a7 row, formed by the helper wave of k_pass while the control wave fits the plane
struct RowPre {
  double X[3][SQ];  // point_this: the scan point in the IMU frame at LiDAR 0's scan end (laserMapping.cpp:660-667)
  double cp[SQ];    // plane weight c_i (:651-656)
  double rw[SQ];    // 1 / R_i after the FIC and the clamp of esekfom.hpp:624-626 (ROWS_DIVIDE: R_i itself)
};
__device__ __forceinline__ void point_phase(const Pass1Args &a, SearchLds &S, int lane, const u32 og[5], int nf, bool &selected,
                                            float4 &pl_out, float &pd2_out, float4 &q_out) {
  selected = false;
  pl_out = make_float4(0.f, 0.f, 0.f, 0.f), pd2_out = 0.f;
  float4 (*const nbp)[SQ] = S.nbp;
  // The control wave issues NO global store before its tile - on gfx9 a wave's stores and loads share one in-order counter
  // (vmcnt), so a store in front of the neighbour gather makes the gather's wait a wait for the store's round trip too. The
  // helper wave stores the per-point state after the barrier (helper_post), from LDS; unit_cov and the trace are its as well.
  if (nf == 5) {  // gate `size < 5 || d2[4] > 5` (:587): only d2 <= 5 candidates were kept
    // ---- esti_plane<float> (common_lib.h:144-190) ----
    float A[5][3];
    {
      float4 m[5];
#ifdef ATTR_NO_GATHER
      for (int k = 0; k < 5; k++) m[k] = make_float4(S.w[lane].x + 0.3f * (float)(k & 1), S.w[lane].y + 0.3f * (float)(k >> 1), S.w[lane].z + 0.01f * (float)k, 0.001f);
#else
#pragma unroll
      for (int k = 0; k < 5; k++) m[k] = a.map_in[og[k]];
#endif
      // all five gathers in flight at once (left to itself the compiler fetched the five normal_y words one after the
      // other, each behind a wait: four extra round trips on this wave's chain)
      asm volatile("" : "+v"(m[0].w), "+v"(m[1].w), "+v"(m[2].w), "+v"(m[3].w), "+v"(m[4].w));
#pragma unroll
      for (int k = 0; k < 5; k++) {
        A[k][0] = m[k].x, A[k][1] = m[k].y, A[k][2] = m[k].z;
        nbp[k][lane] = m[k];
      }
    }
    PH(0, 4);
    float nv[3], pabcd[4];
    PH(0, 5);
    qr_solve_5x3(A, nv);
    PH(0, 6);
    float n = sqrtf(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    pabcd[0] = nv[0] / n, pabcd[1] = nv[1] / n, pabcd[2] = nv[2] / n;
    // (float)(1.0 / (double)n), common_lib.h:180 (`1.0 / n` with n a float promoted to double, stored to float): ONE division of
    // float-representable operands in double, rounded to float - which is the correctly rounded float quotient (double rounding
    // is innocuous for a single +, -, x, / or sqrt when the wide format has >= 2 p + 2 = 50 bits: Figueroa 1995). The float
    // division is a third of the f64 one's instructions on this wave's chain; -DKS_PD_F64 builds the literal form (same bits:
    // tests/test_gpu_parity.py compares either against the oracle's).
#ifdef KS_PD_F64
    pabcd[3] = (float)(1.0 / (double)n);
#else
    pabcd[3] = 1.0f / n;
#endif
    bool plane_ok = true;
#pragma unroll
    for (int k = 0; k < 5; k++) {  // the QR overwrote A: the five points come back from this lane's LDS slots
      const float4 m = nbp[k][lane];
      if (fabsf(pabcd[0] * m.x + pabcd[1] * m.y + pabcd[2] * m.z + pabcd[3]) > a.plane_th) plane_ok = false;
    }
    pl_out = make_float4(pabcd[0], pabcd[1], pabcd[2], pabcd[3]);
    if (plane_ok) {
      float pd2;
      const float4 w = S.w[lane];
      if (residual_gate(pabcd, w.x, w.y, w.z, S.nb[lane], pd2)) {
        selected = true;
        pd2_out = pd2;
      }
    }
  }
  PH(0, 7);
  q_out = S.q[lane];
}


// Phases A .. C for the queries [q0, q0 + 64) n [0, qend) of this workgroup. Returns ROLE_CONTROL in the control wave (with
// its lane's PointOut filled), ROLE_RETIRE in the search waves once they have nothing left to do - and
// ROLE_HELPER in the second wave: it stays to take the plane-independent work off the control wave's chain (helper_pre /
// helper_post below; the caller runs them).
// SKIP: the kernels of a handle with MALIO_OPT_SEARCH_SKIP on - every list walk leaves its certificate, a search pass that is
// not the first of its scan (dy.skip, decided per pass) keeps cached neighbours where the certificate allows (phase A').
// SKIP = false is the search of rounds 1-3 with no trace of any of it.
// KS_QUAD (round 6, built exactly, measured, NOT the default): phase C - plane fit, gates, unit_cov - by the four lanes of every
// query on ALL four waves (quad_fit.hpp: ~470 instructions per lane instead of ~1 000) instead of one lane per query on the
// control wave; what the helper wave did beside the fit without needing the neighbours (traces, point_this, 1 / R_i) done beside
// phase A, when it used to wait. Bit-identical (the whole GPU suite passes on it; tests/test_quad_fit.py pins the lane-parallel
// fit to the oracle on the host) - and SLOWER wherever the GPU is full: k_pass 27.0 -> 31.0 us at BASELINE config 2, 36.0 -> 39.5
// at config 5, 23.3 -> 24.8 at config 3, even at config 1 (profiles/round6/r06e_quad_phase_c.txt). One lane per query is the
// form that spends the fewest WAVE-instructions on a fit; four lanes replicate row 0, the norms, the pivots and every scalar of
// the step, and what the chain of one workgroup gains (~560 instructions) the six other workgroups of the CU pay for in issue
// slots (+1 900 wave-instructions per workgroup on SIMDs that are busy half the time): the pass is bound by VALU issue about as
// much as by its L1 queue. 0 (default): phase C of rounds 1-5.
#ifndef KS_QUAD
#define KS_QUAD 0
#endif
#ifndef KS_REUSE_ROWS
#define KS_REUSE_ROWS 1  // reuse passes that may speculate run as k_reuse_rows -> k_final_reduce<4> (0: k_pass' reuse form, rounds 3-5)
#endif
constexpr int ROLE_RETIRE = 0, ROLE_CONTROL = 1, ROLE_HELPER = 2;
// PIPE2: the level-2 walk is the pipelined one (nl_search; not in the device loop's k_search<true, .>, whose extra reuse branch
// makes the register allocator spill with it)
// pre_a(lane, i, in_range) (KS_QUAD): called by the HELPER wave during phase A, lane = query - the plane-independent work that
// needs nothing from the search. cp_of(ucov) (KS_QUAD, k_pass): the plane weight c_i under the guessed extrema, or null.
struct NoPreA {
  __device__ __forceinline__ void operator()(int, int, bool) const {}
};
struct NoCp {
  static constexpr bool enabled = false;
  __device__ __forceinline__ double operator()(double) const { return 0.0; }
};
template <bool DEV, bool SKIP, bool PIPE2, class PreA = NoPreA, class CpOf = NoCp>
__device__ __forceinline__ int search_wg(const Pass1Args &a, const NlView &nl1, const NlView &nl2, const QuatConst &qc,
                                         const PassDyn &dy, SearchLds &S, int q0, int qend, PointOut &po, PreA pre_a = PreA(),
                                         CpOf cp_of = CpOf(), double *cp_out = nullptr) {
  auto qidx = [&](int l) { return q0 + l; };
  const int lane_ = (int)(threadIdx.x & 63);
  const bool cwave = (int)(threadIdx.x >> 6) == 0;
  po.selected = false, po.ucov = 0.0, po.tr = 0.0, po.pd2 = 0.f;
  po.pl = make_float4(0.f, 0.f, 0.f, 0.f), po.q = po.pl;
  // ---- phase A ----
  const int i = qidx(lane_);  // meaningful for the control wave only
  bool mine = cwave && i < qend;
  bool keep = false;
  PH(0, 0);
  PH_ENTER();
  if (blockIdx.x == 0 && threadIdx.x < MM_SLOTS) mm_reset_slot(dy.mm_next, threadIdx.x);  // the OTHER parity's slots, for the next pass
  if (cwave) {
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    uint4 pcv = make_uint4(~0u, ~0u, 0u, 0u);  // the point's cached directory probe; ~0: none (no real key has bit 63 set)
    if (mine) {
      const float4 q = a.scan[i];
      // the lazy normal_y commit of the previous pass (commit_normal_y), done here: its two loads travel with the scan
      // point's instead of opening phase C with two dependent round trips
      unsigned char sel_prev = 0;
      double trace_prev = 0.0;
      if (dy.commit_prev) sel_prev = a.sel[i], trace_prev = a.trace[i];
      // The level-1 directory probe of the point's LAST search pass (round 5): between two search passes of one update the
      // iterate moves by centimetres and 99 % of the points stay in their cell - their list is where it was. 16 bytes per
      // point in one coalesced load here, against a dependent round trip to a line of the directory of its own per query
      // in phase B (3 - 4 us of that phase under load, a fifth of its lines).
      if (KS_PCACHE && (dy.skip & 2)) pcv = a.pcache[i];
      double nb;
      world_point(qc, q, __float_as_int(q.w) & 0xFF, w.x, w.y, w.z, nb);
      // (helper_post stores it, see point_phase - except on a tile shard, where a point of another shard loses its world
      // point below and is nobody's to store later)
      if (a.part.world > 1) STATE_ST(a.world4[i] = w;)
      S.nb[lane_] = sqrt(nb);  // (sqrt(p_body.norm()), :599: formed here, off phase C's chain)
      S.q[lane_] = q;
      if (!part_owns(a.part, w.x, w.y, w.z)) {  // another shard serves this point (same bits there: same decision)
        mine = false;
        a.nfound[i] = NF_NOTMINE, a.sel[i] = 0;
        w = make_float4(3e9f, 3e9f, 3e9f, 0.f);  // far from every list: its search lanes find an empty cell
      } else if (dy.commit_prev && !(sel_prev && !a.extrinsic_est_en)) {
        a.ny[i] = (float)trace_prev;
      }
    }
    S.w[lane_] = w;
    // (EVERY lane of the tile: a lane past the end of its LiDAR segment searches at (0, 0, 0) like any other, and whatever an earlier
    // workgroup - of another handle, another map - left in this LDS must not pass for its cached probe: it did, in scenes around the
    // origin, and the walk of a list that is not there faulted)
    S.pc[lane_] = pcv;
    // ---- phase A': may this query keep the neighbours of its last walk? (laserMapping.cpp:582-591 searches every point
    // again whenever ekfom_data.converge is set; between two such passes of one update the iterate moves by centimetres.)
    // The last walk of the point, at w0, left a radius r0 inside which there is no map point but the cached ones
    // (nl_search's lb2). Every outsider is therefore farther than r = r0 - |w - w0| from the new point w. If that beats
    // the fifth cached distance - or the acceptance radius sqrt 5 (:587), when fewer than five cached points lie inside
    // it - the search would return exactly the cached points, ranked by their NEW distances under the order (d2, index):
    // those are recomputed here the way the search computes them (ikd_Tree.cpp:1697, no FMA) and inserted into an empty
    // list. Same set, same order, same bits; anything the bound cannot decide walks the lists as before.
    if (SKIP && mine && (dy.skip & 1)) {
      const float4 ce = a.cert[i];
      const int nfo = a.nfound[i];
      if (nfo <= 5 && ce.w > 0.f) {
        Top5 t;
#pragma unroll
        for (int k = 0; k < 5; k++) t.k[k] = TOP5_MAXKEY;
        u32 id[5];
#pragma unroll
        for (int k = 0; k < 5; k++) id[k] = a.nbr[(size_t)k * a.N + i];
#pragma unroll
        for (int k = 0; k < 5; k++) {
          u64 key = TOP5_MAXKEY;
          if (id[k] != INVALID) {
            const float4 m = a.map_in[id[k]];
            const float ddx = w.x - m.x, ddy = w.y - m.y, ddz = w.z - m.z;
            key = top5_key(ddx * ddx + ddy * ddy + ddz * ddz, id[k]);
          }
          (void)top5_insert(t, key);
        }
        const float ex = w.x - ce.x, ey = w.y - ce.y, ez = w.z - ce.z;
        const float moved = sqrtf(ex * ex + ey * ey + ez * ez) * 1.00001f;
        const float r = (ce.w - moved) * 0.9999f;  // (the slack dwarfs every rounding above: 3e-7 relative)
        const float lim = __uint_as_float(__float_as_uint(5.0f) + 1u);
        float need = lim;  // fewer than five inside the radius: no outsider may come inside sqrt 5
        if (t.og(4) != INVALID && t.d(4) < lim) need = t.d(4);
        keep = r > 0.f && r * r > need;
        if (keep) {
          int nf = 0;
#pragma unroll
          for (int k = 0; k < 5; k++) {
            const bool in = t.og(k) != INVALID && t.d(k) <= 5.0f;
            S.og[k][lane_] = in ? t.og(k) : INVALID;
            nf += in ? 1 : 0;
          }
          S.nf[lane_] = (unsigned char)nf;
        }
      }
    }
    if (SKIP) {
      S.keep[lane_] = keep ? 1 : 0;
      if (i < qend) a.kept[i] = keep ? 1 : 0;
    }
  }
#if KS_QUAD
  if ((int)(threadIdx.x >> 6) == 1) pre_a(lane_, i, i < qend);  // (the helper wave: it waited here until round 6)
#endif
  // bit 0: a point of this workgroup is served here; bit 1: one of them has to walk the lists (only the control wave knows;
  // __syncthreads_or would reduce !!predicate, not the bits)
  int flags = 3;
  if (SKIP) {
    if (cwave) {
      const int fl = (__ballot(mine) ? 1 : 0) | (__ballot(mine && !keep) ? 2 : 0);
      if (lane_ == 0) S.flags = fl;
    }
    __syncthreads();
    flags = __builtin_amdgcn_readfirstlane(S.flags);  // (workgroup-uniform: a scalar)
  } else if (a.part.world > 1) {
    flags = __syncthreads_or(mine ? 1 : 0) ? 3 : 0;
  } else {
    __syncthreads();
  }
  if (a.part.world > 1 && !(flags & 1)) {      // a workgroup of somebody else's tiles
    // Nobody probes here, so nobody would overwrite these points' cached probes: entries of an earlier scan - another map epoch -
    // would pass for this scan's as soon as one of the points crosses into an owned tile with a matching cell key. They are
    // marked empty instead (a lane of a workgroup that does search overwrites its entry in phase B, owned or not).
#ifndef KS_NO_PCACHE_MARK  // (A/B switch of tests/test_partition.py::test_tile_shard_forgets_cached_probes_of_an_earlier_scan: the kernel before the fix)
    if (KS_PCACHE && cwave && (dy.skip & 4) && i < qend) a.pcache[i] = make_uint4(~0u, ~0u, 0u, 0u);
#endif
    po.selected = false, po.skipped = true;  // (k_pass still owes the summation tree a zero tile)
    return cwave ? ROLE_CONTROL : ROLE_RETIRE;
  }
  PH(0, 1);
  if (flags & 2) {
  // ---- phase B ----
  {
    const int ql = threadIdx.x / NL1_G, sub = threadIdx.x % NL1_G;
    if (!SKIP || !S.keep[ql]) {  // (the NL1_G lanes of a query branch together: the shuffles of the merge stay inside the group)
      const float4 ww = S.w[ql];
      Top5 t;
      float lb2;
      u32 st_, cn_;
      if (KS_PCACHE) {
        const uint4 pcv = S.pc[ql];
        const u64 key = cell_key_d((int)floorf(ww.x * nl1.inv_cf), (int)floorf(ww.y * nl1.inv_cf), (int)floorf(ww.z * nl1.inv_cf));  // == nl_probe's
        if (pcv.x == (u32)key && pcv.y == (u32)(key >> 32)) {
          st_ = pcv.z, cn_ = pcv.w;
        } else {
          nl_probe(nl1, ww.x, ww.y, ww.z, st_, cn_);
          if (sub == 0 && (dy.skip & 4) && qidx(ql) < qend) a.pcache[qidx(ql)] = make_uint4((u32)key, (u32)(key >> 32), st_, cn_);
        }
      } else {
        nl_probe(nl1, ww.x, ww.y, ww.z, st_, cn_);
      }
#ifdef MALIO_PHASE_CLOCK
      asm volatile("" ::"v"(st_), "v"(cn_));
      PH_WAVE(8);
#endif
      const bool certified = nl_walk<NL1_G, SKIP, false, KS_EARLY != 0>(nl1, ww.x, ww.y, ww.z, sub, 5.0f, st_, cn_, t, lb2);
#ifdef MALIO_PHASE_CLOCK
      PH_WAVE(4);
#endif
      if (sub == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++) S.og[k][ql] = t.og(k);
        S.nf[ql] = certified ? 5 : NF_PENDING;
        if (SKIP) S.cr[ql] = cert_radius(lb2);
      }
    }
  }
  __syncthreads();
  PH(0, 2);
  PH_NOTE(2, wall_clock64());
  PH_NOTE(3, 0);
  // ---- level 2 (phase B'): the queries level 1 could not certify walk their ~180..900-point level-2 list ----
  {
    const int lane = threadIdx.x & 63;
    // lane <-> query, the same in every wave (a point of another shard sits at 3e9 and is never searched further)
    const bool pend = (qidx(lane) < qend) && S.nf[lane] == NF_PENDING && S.w[lane].x < 1e9f;
    unsigned long long todo = __ballot(pend);
    if (todo) {  // workgroup-uniform: nothing writes S.nf between the barrier above and the one below
      // every wave must have taken its snapshot of the flags before any group rewrites one (the serving group stores the
      // final count): a wave that read S.nf late would see a different `todo`, the assignment of queries to groups below
      // would differ between waves and a query could be left NF_PENDING
      __syncthreads();
      const int npend = __popcll(todo);
      PH_NOTE(3, npend);
      {
        // L2G lanes per uncertified query: the workgroup's 256 / L2G groups walk as many level-2 lists side by side - with
        // the four lanes of level 1 ALL of a workgroup's uncertified queries in one round. Rounds 1-3 gave such a query a
        // whole wave (a walk of two batches, but four queries at a time: a workgroup of the tunnel scene, BASELINE config 5,
        // holds 9..64 of them and walked up to six times in a row) and handed workgroups with more than 24 to a kernel of
        // their own, k_search_tail, behind the search (22 us), which also kept such passes off the one-kernel path.
        // Swept at config 5 (profiles/round4/r04c_level2_groups.txt), whole search pass: 64 lanes + tail 84 us, 16 lanes + tail
        // 75, and without any deferral 32 / 16 / 8 / 4 lanes -> 68 / 57 / 50 / 49 us; configs 1-4 do not move.
        constexpr int L2G = KS_L2G, NGRP = KS_BLK / L2G;
        const int grp = (int)threadIdx.x / L2G, sub = (int)threadIdx.x % L2G;
        for (int r0 = 0; r0 < npend; r0 += NGRP) {
          const int ord = r0 + grp;
          if (ord < npend) {  // (the lanes of a group branch together: the shuffles of the merge stay inside it)
            const int l = nth_set_bit(todo, ord);
            const float4 ww = S.w[l];
            Top5 t;
            float lb2;
            nl_search<L2G, SKIP, PIPE2>(nl2, ww.x, ww.y, ww.z, sub, 5.0f, t, lb2);  // merged list is identical in every lane of the group
            if (sub == 0) {
              int nf = 0;
#pragma unroll
              for (int k = 0; k < 5; k++) S.og[k][l] = t.og(k), nf += (t.og(k) != INVALID);
              S.nf[l] = (unsigned char)nf;
              if (SKIP) S.cr[l] = cert_radius(lb2);
            }
          }
        }
      }
      __syncthreads();
    }
  }
  }  // (somebody walks)
#if KS_QUAD
  // ---- phase C' (all four waves, the four lanes of a query together): a3 + gates + plane_cov ----
  {
    int tq = (int)threadIdx.x;
    asm volatile("" : "+v"(tq));  // (query and lane are formed again: shared with phase B's they stay live across the list walk, which has no register for them)
    const int ql = tq / 4, sub = tq % 4;
    const int iq = qidx(ql);
    const float4 wq = S.w[ql];
    const int nfq = S.nf[ql];
    const bool fit = iq < qend && wq.x < 1e9f && nfq == 5;  // (the four lanes agree) gate `size < 5 || d2[4] > 5` (:587)
    bool selected = false;
    float4 pl = make_float4(0.f, 0.f, 0.f, 0.f);
    float pd2v = 0.f;
    double ucov = 0.0;
    if (fit) {
      // neighbour 0 in all four lanes (one address: one request), neighbour sub + 1 in each
      const u32 o0 = S.og[0][ql], om = S.og[sub + 1][ql];
#ifdef ATTR_NO_GATHER
      const float4 m0 = make_float4(wq.x, wq.y, wq.z, 0.001f), mm = make_float4(wq.x + 0.3f * (float)(sub & 1), wq.y + 0.3f * (float)(sub >> 1), wq.z + 0.01f * (float)sub, 0.001f);
#else
      const float4 m0 = a.map_in[o0], mm = a.map_in[om];
#endif
      const float row0[3] = {m0.x, m0.y, m0.z}, mine[3] = {mm.x, mm.y, mm.z};
      float pabcd[4];
      const bool plane_ok = quad::esti_plane_quad<float>(row0, mine, a.plane_th, pabcd);
      ucov = quad::unit_cov_quad<float, double>(a.cov_threshold, m0.w, mm.w);
      pl = make_float4(pabcd[0], pabcd[1], pabcd[2], pabcd[3]);
      float pd2;
      if (plane_ok && residual_gate(pabcd, wq.x, wq.y, wq.z, S.nb[ql], pd2)) selected = true, pd2v = pd2;
    }
    if (sub == 0) {
      S.selc[ql] = selected ? 1 : 0, S.plc[ql] = pl, S.pd2c[ql] = pd2v, S.ucv[ql] = ucov;
      if (CpOf::enabled) cp_out[ql] = cp_of(ucov);
    }
  }
  __syncthreads();
  {
    // (wave and lane are formed again: kept across the list walk they are what the register allocator spills)
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int wv = tid >> 6, ln = tid & 63;
    if (wv >= 2) return ROLE_RETIRE;
    if (wv == 1) return ROLE_HELPER;
    po.selected = S.selc[ln] != 0, po.pl = S.plc[ln], po.pd2 = S.pd2c[ln], po.q = S.q[ln];
  }
  PH(0, 8);
  return ROLE_CONTROL;
#endif
  if (!cwave) {
    if ((int)(threadIdx.x >> 6) == 1) return ROLE_HELPER;
    return ROLE_RETIRE;
  }
  // ---- phase C (control wave) ----
  // From here to its tile this wave is ONE dependent chain of ~3 000 instructions, and its workgroup leaves when the chain ends.
  // (Raised wave priority for it - s_setprio 1 .. 3, the helper wave likewise - was measured in round 6 and changes nothing:
  // profiles/round6/r06b_wave_priority.txt. The chain's length is its instructions, not the slots it loses.)
  int lane = lane_;
  asm volatile("" : "+v"(lane));  // the query index is formed again from here on: kept across the list walk it is the one
  const int ic = qidx(lane);       // value the register allocator spills (8 B of scratch per lane for a 32-bit add)
  u32 og[5];
#pragma unroll
  for (int k = 0; k < 5; k++) og[k] = S.og[k][lane];
  const int nf = S.nf[lane];
  PH(0, 3);
  // (what phase A knew is read back from LDS instead of being kept in registers across the list walk: the walk has none to spare)
  const float4 wq = S.w[lane];
  const bool served = ic < qend && wq.x < 1e9f;  // == mine: a point of another shard sits at 3e9
  if (served)
    point_phase(a, S, lane, og, nf, po.selected, po.pl, po.pd2, po.q);
  PH(0, 8);
  return ROLE_CONTROL;
}

// ---- the helper wave (round 5) ---------------------------------------------------------------------------------------------
// The control wave's phase C is one dependent chain (gather -> QR -> gates -> ... -> row -> 16 MFMAs), 9.5 us of a 21 us
// workgroup at BASELINE config 2, and three waves used to retire beside it. Everything on that chain that does not need the
// PLANE is now the second wave's, lane = query as well:
//   helper_pre   (while the control wave gathers and fits): the five neighbours' normal_y -> unit_cov (esti_plane's
//                plane_cov, common_lib.h:159-173); trace(Sigma_p) under both clamp rules (trace_both); for k_pass the row's
//                plane-independent factors - point_this (laserMapping.cpp:660-667), the plane weight c_i (:651-656) and
//                1 / R_i after the FIC (:716-721, esekfom.hpp:624-626, for the accepted-point trace: only accepted points
//                have rows) - left in LDS;
//   ONE barrier  the control wave arrives with the accept flags in LDS;
//   helper_post  picks the trace, stores unit_cov and the trace, reduces the extrema of the wave's 64 points and publishes
//                them (a4) - while the control wave forms the plane-dependent rest of the row and the tile.
// Every quantity is formed by the same expression on the same operands as before (one function, two callers): same bits.
// i: the lane's point (sorted index), served: it is this workgroup's to serve; nf: its neighbour count.
__device__ __forceinline__ void helper_unit_cov_trace(const Pass1Args &a, SearchLds &S, int lane, int i, bool served, int nf,
                                                      const u32 og[5], const float4 q, double &ucov, double &trS) {
  ucov = 0.0, trS = 0.0;
  double trR = 0.0;
  if (served) {
    if (nf == 5) {
      float W[5];
#pragma unroll
      for (int k = 0; k < 5; k++)
#ifdef ATTR_NO_GATHER
        W[k] = 0.001f;
#else
        W[k] = a.map_in[og[k]].w;
#endif
      ucov = plane_unit_cov(a.cov_threshold, W);
    }
    const int packed = __float_as_int(q.w);
    trace_both(a, q, packed & 0xFF, packed >> 8, trS, trR);
  }
  S.ucv[lane] = ucov, S.trS[lane] = trS, S.trR[lane] = trR;
}
// after the barrier, SEARCH pass: flag -> trace; the point's whole per-point state (the control wave stores none of it:
// see point_phase) - world point, certificate, neighbour ids and count, plane, unit_cov, residual, flag, trace -; a4
template <bool SKIP>
__device__ __forceinline__ void helper_post(const Pass1Args &a, u64 *mm_cur, SearchLds &S, int lane, int i, bool served, int nf) {
  const bool selected = served && S.selc[lane] != 0;
  const double ucov = S.ucv[lane];
  const double tr = selected ? S.trS[lane] : S.trR[lane];
  if (served) {
    const float4 w = S.w[lane];
    if (a.part.world <= 1) STATE_ST(a.world4[i] = w;)  // (a tile shard: phase A stored it)
    if (SKIP && !S.keep[lane]) STATE_ST(a.cert[i] = make_float4(w.x, w.y, w.z, S.cr[lane]);)  // this pass walked: its certificate
#pragma unroll
    for (int k = 0; k < 5; k++) STATE_ST(a.nbr[(size_t)k * a.N + i] = S.og[k][lane];)
    STATE_ST(a.nfound[i] = (unsigned char)nf;)  // (feats_down_world of a search pass is world4: malio_scan_get reads it from there)
    if (nf == 5) {
      STATE_ST(a.plane[i] = S.plc[lane];)
      STATE_ST(a.ucov[i] = ucov;)
    }
    if (selected) STATE_ST(a.pd2[i] = S.pd2c[lane];)
    STATE_ST(a.sel[i] = selected ? 1 : 0;)
    STATE_ST(a.trace[i] = tr;)
  }
  wave_minmax_publish(a, mm_cur, selected, ucov, tr);
}
// ... REUSE pass (k_pass): world point, the previous pass' lazy normal_y fold (commit_normal_y, from the flag and trace the
// helper read BEFORE the barrier), residual, flag, trace; a4
__device__ __forceinline__ void helper_post_reuse(const Pass1Args &a, u64 *mm_cur, SearchLds &S, int lane, int i, bool served,
                                                  int commit_prev, unsigned char sel_old, double trace_old) {
  const bool selected = served && S.selc[lane] != 0;
  const double ucov = S.ucv[lane];
  const double tr = selected ? S.trS[lane] : S.trR[lane];
  if (served) {
    const float4 w = S.w[lane];
    a.world[i] = w.x, a.world[a.N + i] = w.y, a.world[2 * a.N + i] = w.z;
    if (commit_prev && !(sel_old && !a.extrinsic_est_en)) a.ny[i] = (float)trace_old;
    if (selected) a.pd2[i] = S.pd2c[lane];
    a.sel[i] = selected ? 1 : 0;
    a.trace[i] = tr;
  }
  wave_minmax_publish(a, mm_cur, selected, ucov, tr);
}

// DEV = true: one pass of the device-resident update loop (DevLoop): exits when the loop is over, runs the REUSE pass on
// its first wave when the control block asks for one (a reuse pass then costs one launch of this grid, no second
// kernel that would have to be enqueued and skipped), and takes state, parities and commit_prev from the block.
template <bool DEV, bool SKIP>
__global__ void __launch_bounds__(KS_BLK) __attribute__((amdgpu_waves_per_eu(KS_WPE, KS_WPE))) k_search(Pass1Args a, NlView nl1, NlView nl2) {
  __shared__ SearchLds S;
  __shared__ float4 s_nbp[5][SQ];
  POISON_LDS(S);
  POISON_LDS(s_nbp);
  POISON_SYNC();
  if (threadIdx.x == 0) S.nbp = s_nbp;  // (read by the control wave in phase C, behind the barriers of phases A and B)
  if (DEV && a.dl->done) return;
  const QuatConst &qc = DEV ? a.dl->qc : a.qc;
  const PassDyn dy = pass_dyn<DEV>(a);
  if (DEV && !a.dl->converge) {  // REUSE pass: thread = point, so the first quarter of the grid does all of it
    const int base = (int)(blockIdx.x * KS_BLK);
    if (base >= a.N) return;
    bool selected;
    double ucov, tr;
    float4 pl, q;
    float pd2;
    reuse_point(a, qc, dy.commit_prev, base + (int)threadIdx.x, selected, ucov, tr, pl, pd2, q);
    wave_minmax_publish(a, dy.mm_cur, selected, ucov, tr);
    if (blockIdx.x == 0 && threadIdx.x < MM_SLOTS) mm_reset_slot(dy.mm_next, threadIdx.x);
    return;
  }
  PointOut po;
#if KS_QUAD
  // (phase A's shadow, helper wave, lane = query: the trace under both clamp rules - it needs the scan point, not the search)
  auto pre_a = [&](int lane, int i, bool in) {
    double trS = 0.0, trR = 0.0;
    if (in) {
      const float4 q = a.scan[i];
      const int packed = __float_as_int(q.w);
      trace_both(a, q, packed & 0xFF, packed >> 8, trS, trR);
    }
    S.trS[lane] = trS, S.trR[lane] = trR;
  };
  const int role = search_wg<DEV, SKIP, (KS_PIPE != 0) && !DEV>(a, nl1, nl2, qc, dy, S, (int)blockIdx.x * SQ, a.N, po, pre_a);
  if (role == ROLE_RETIRE || po.skipped) return;
  if (role == ROLE_HELPER) {  // (phase C' left flag, plane, residual and unit_cov in LDS; its barrier is behind us)
    const int lane = (int)(threadIdx.x & 63), i = (int)blockIdx.x * SQ + lane;
    helper_post<SKIP>(a, dy.mm_cur, S, lane, i, i < a.N && S.w[lane].x < 1e9f, S.nf[lane]);
  }
  PH(0, 9);
  PH_EXIT();
  return;
#else
  const int role = search_wg<DEV, SKIP, (KS_PIPE != 0) && !DEV>(a, nl1, nl2, qc, dy, S, (int)blockIdx.x * SQ, a.N, po);
  if (role == ROLE_RETIRE || po.skipped) return;
#endif
  {
    const int lane = (int)(threadIdx.x & 63), i = (int)blockIdx.x * SQ + lane;
    if (role == ROLE_HELPER) {  // (see helper_unit_cov_trace)
      const bool served = i < a.N && S.w[lane].x < 1e9f;
      const int nf = S.nf[lane];
      u32 og[5];
#pragma unroll
      for (int k = 0; k < 5; k++) og[k] = S.og[k][lane];
      double ucov, trS;
      helper_unit_cov_trace(a, S, lane, i, served, nf, og, S.q[lane], ucov, trS);
      __syncthreads();
      helper_post<SKIP>(a, dy.mm_cur, S, lane, i, served, nf);  // per-point state; a4 over this workgroup's 64 queries
      return;
    }
    S.selc[lane] = po.selected ? 1 : 0, S.plc[lane] = po.pl, S.pd2c[lane] = po.pd2;
    __syncthreads();
  }
  PH(0, 9);
  PH_EXIT();
}

// REUSE pass of the host-driven path, thread per point (the device loop runs it inside k_search<true>).
__global__ void __launch_bounds__(BLK) k_reuse(Pass1Args a) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  const PassDyn dy = pass_dyn<false>(a);
  bool selected;
  double ucov, tr;
  float4 pl, q;
  float pd2;
  reuse_point(a, a.qc, dy.commit_prev, i, selected, ucov, tr, pl, pd2, q);
  block_minmax(a, dy, selected, ucov, tr);
}

__global__ void k_mm_init(u64 *slots) { mm_reset_slot(slots, threadIdx.x); }

// Staged (multi-GPU) path: one wave folds the slots into [max_u, -min_u, max_R, -min_R, M] for the all-reduce.
__global__ void __launch_bounds__(64) k_minmax_reduce(const u64 *__restrict__ slots, int extrinsic_est_en, double *out) {
  double o5[5];
  mm_fold_wave(slots, extrinsic_est_en, o5);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 5; k++) out[k] = o5[k];
    out[5] = 0.0;  // (word 5 of the extrema row: reserved - it carried the deferral score of rounds 1-3)
  }
}

// ------------------------------------------------------------------------------------------------
struct Pass2Args {
  int N, L, extrinsic_est_en;
  const float4 *scan;
  const float4 *plane;
  const float *pd2;
  const double *ucov;
  const double *trace;
  const unsigned char *sel;
  int seg_block0[MALIO_MAX_LIDAR + 1];  // first workgroup of each LiDAR segment
  int seg_start[MALIO_MAX_LIDAR + 1];   // first sorted point of each LiDAR segment
  PassConst pc;
  WeightConst wc;
  const double *minmax4;  // [max_ucov, -min_ucov, max_R, -min_R] when the caller reduced them (multi-GPU), else null
  const u64 *mmslots;  // extrema slots of stage 1 (single-GPU path: folded here by the first wave)
  double *mm_out;         // where workgroup 0 publishes the folded extrema + M for the host
  double *partials;       // [NSUM][pstride]: entry-major, so the final sum reads each entry's partials contiguously
  int pstride;
  double *rows;           // optional [N][14]: u[12], hs, r   (sorted order)
  // device loop (DEV = true): the matrix form of the state and the slot parity come from *dl
  const DevLoop *dl;
  const u64 *mm_base;     // [2 parities][MM_SLOTS][5]
};

// a5 + a7: weights and the 12 non-zero entries of the (c_i-scaled) Jacobian row of one accepted point
// what point_row reads of one point (loaded before the workgroup waits for the extrema, see k_rows_reduce)
struct RowIn {
  float4 q, pl;
  double ucov, trace;
  float pd2;
};
// (in pieces, so that k_pass' helper wave can form the plane-independent ones - row_plane_weight, row_point_imu,
// row_point_noise - while the control wave still fits the plane: one expression per quantity whoever evaluates it)
// plane weight c_i (laserMapping.cpp:651-656)
__device__ __forceinline__ double row_plane_weight(const WeightConst &wc, const double mm[4], double ucov) {
  const double max_u = mm[0], min_u = -mm[1];
  double cp = ucov;
  if (cp == 0)
    cp = 1;
  else if (max_u == min_u)
    cp = (wc.plane_cov_max + wc.plane_cov_min) / 2;
  else
    cp = 1 / ((wc.plane_cov_max - wc.plane_cov_min) * (cp - min_u) / (max_u - min_u) + wc.plane_cov_min);
  return cp;
}
// point_this: q0 * p_be + t0, the point in the IMU frame at LiDAR 0's scan end (:658-667)
__device__ __forceinline__ D3 row_point_imu(const PassConst &pc, int lid, D3 p) {
  const LidarConst &lc = pc.lid[lid];
  if (lid == 0) return mulR(pc.R0, p) + D3{pc.t0[0], pc.t0[1], pc.t0[2]};
  D3 y = mulR(lc.Rl, p) + D3{lc.tl[0], lc.tl[1], lc.tl[2]};
  return mulR(lc.Rtc, y) + D3{lc.ttc[0], lc.ttc[1], lc.ttc[2]};
}
// point noise R_i by FIC (:716-721)
__device__ __forceinline__ double row_point_noise(const WeightConst &wc, int extrinsic_est_en, const double mm[4], double trace) {
  const double max_c = mm[2], min_c = -mm[3];
  double R = extrinsic_est_en ? trace : 0.0;
  const double lo = min_c + (max_c - min_c) * wc.range_min, hi = min_c + (max_c - min_c) * wc.range_max;
  if (R < lo)
    R = wc.point_cov_min;
  else if (R > hi)
    R = wc.point_cov_max;
  else
    R = (wc.point_cov_max - wc.point_cov_min) * (R - lo) / ((wc.range_max - wc.range_min) * (max_c - min_c)) +
        wc.point_cov_min;
  return R;
}
// the plane-dependent rest: the 12 non-zero entries of the c_i-scaled row and its residual (:676-693,707,714-715)
__device__ __forceinline__ void row_finish(int extrinsic_est_en, const PassConst &pc, int lid, const float4 q, const float4 pl,
                                           float pd2, D3 X, double cp, double u[12], double &hs) {
  D3 p{(double)q.x, (double)q.y, (double)q.z};
  const LidarConst &lc = pc.lid[lid];
  D3 n{(double)pl.x, (double)pl.y, (double)pl.z};
  D3 Cv = mulRt(pc.Rw, n);  // s.rot.conjugate() * norm_vec (:676)
  D3 A = cross(X, Cv);        // point_crossmat * C (:677)
  D3 B{0, 0, 0}, Cb{0, 0, 0};
  if (extrinsic_est_en) {
    if (lid == 0) {
      B = cross(p, mulRt(pc.R0, Cv));  // :684 (p_be == p for LiDAR 0)
      Cb = Cv;
    } else {
      Cb = mulRt(lc.Rtc, Cv);            // :689
      B = cross(p, mulRt(lc.Rl, Cb));    // :690
    }
  }
  u[0] = n.x * cp, u[1] = n.y * cp, u[2] = n.z * cp;  // row * cov_plane[i] (:714)
  u[3] = A.x * cp, u[4] = A.y * cp, u[5] = A.z * cp;
  u[6] = B.x * cp, u[7] = B.y * cp, u[8] = B.z * cp;
  u[9] = Cb.x * cp, u[10] = Cb.y * cp, u[11] = Cb.z * cp;
  hs = (-1.0) * (double)pd2 * cp;  // :707,715
}
__device__ __forceinline__ void point_row(const WeightConst &wc, int extrinsic_est_en, const PassConst &pc, const double mm[4],
                                          const RowIn &in, int lid, double u[12], double &hs, double &r) {
  const double cp = row_plane_weight(wc, mm, in.ucov);
  const D3 X = row_point_imu(pc, lid, D3{(double)in.q.x, (double)in.q.y, (double)in.q.z});
  row_finish(extrinsic_est_en, pc, lid, in.q, in.pl, in.pd2, X, cp, u, hs);
  r = row_point_noise(wc, extrinsic_est_en, mm, in.trace);
}

// One workgroup = 256 consecutive sorted points of ONE LiDAR. Rows go to LDS, each wave turns its 64 rows into a
// 16x16 block of sums with 16 f64 MFMAs, the 4 wave blocks are added in a fixed order.
template <bool DEV>
__global__ void __launch_bounds__(BLK) k_rows_reduce(Pass2Args a) {
  __shared__ double SA[BLK][17];  // a_p: u / r (r clamped as esekfom.hpp:624-626), u[0..2], 0   (+1 pad)
  __shared__ double SB[BLK][17];  // b_p: u, hs, 0 0 0                                           (+1 pad)
  __shared__ double DW[BLK / 64][16][16];
  __shared__ double mm_s[5];
  POISON_LDS(SA);
  POISON_LDS(SB);
  POISON_LDS(DW);
  POISON_LDS(mm_s);
  POISON_SYNC();
  if (DEV && a.dl->done) return;
  const u64 *mmslots = DEV ? a.mm_base + (size_t)a.dl->mm_parity * MM_SLOTS * 5 : a.mmslots;
  const PassConst &pc = DEV ? a.dl->pc : a.pc;
  PH(1, 0);
  // this thread's point: its loads are issued before the extrema fold and the barrier behind it (nothing below needs
  // them until then: a kernel this short is one memory round trip deep, and this puts it under the fold)
  int lid = 0;
#pragma unroll
  for (int l = 1; l < MALIO_MAX_LIDAR; l++)
    if (l < a.L && (int)blockIdx.x >= a.seg_block0[l]) lid = l;
  const int i = a.seg_start[lid] + ((int)blockIdx.x - a.seg_block0[lid]) * BLK + threadIdx.x;
  const bool in = i < a.seg_start[lid + 1];
  const bool selected = in && a.sel[i] != 0;
  RowIn rin;
  rin.q = make_float4(0.f, 0.f, 0.f, 0.f), rin.pl = rin.q, rin.ucov = 0, rin.trace = 0, rin.pd2 = 0.f;
  if (selected) rin.q = a.scan[i], rin.pl = a.plane[i], rin.ucov = a.ucov[i], rin.trace = a.trace[i], rin.pd2 = a.pd2[i];
  // ---- a4 fold: the first wave of every workgroup folds the 64 extrema slots of stage 1 (2.5 KB from L2) ----
  if (a.minmax4) {
    if (threadIdx.x < 4) mm_s[threadIdx.x] = a.minmax4[threadIdx.x];
    if (blockIdx.x == 0 && a.mm_out && threadIdx.x < 64) {
      // staged path with the extrema supplied by the caller (a guess, or all-reduced): this shard's own extrema are
      // still wanted - the caller verifies its guess against them - and are folded here instead of by a separate launch
      double o5[5];
      mm_fold_wave(mmslots, a.extrinsic_est_en, o5);
      if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++) a.mm_out[k] = o5[k];
        a.mm_out[5] = 0.0;
      }
    }
  } else if (threadIdx.x < 64) {
    double o5[5];
    mm_fold_wave(mmslots, a.extrinsic_est_en, o5);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) mm_s[k] = o5[k];
      if (blockIdx.x == 0 && a.mm_out) {
#pragma unroll
        for (int k = 0; k < 5; k++) a.mm_out[k] = o5[k];
        a.mm_out[5] = 0.0;
      }
    }
  }
  __syncthreads();
  PH(1, 1);
  const double mm[4] = {mm_s[0], mm_s[1], mm_s[2], mm_s[3]};
  double u[12], hs = 0, r = 1;
#pragma unroll
  for (int k = 0; k < 12; k++) u[k] = 0;
  if (selected) point_row(a.wc, a.extrinsic_est_en, pc, mm, rin, lid, u, hs, r);
  PH(1, 2);
  if (a.rows && in) {
    double *row = a.rows + (size_t)i * 14;
#pragma unroll
    for (int k = 0; k < 12; k++) row[k] = u[k];
    row[12] = hs;
    row[13] = selected ? r : 0.0;
  }
  double rc = r;
  if (rc < 0.0001) rc = 0.001;  // esekfom.hpp:624-626
  // One reciprocal instead of 12 f64 divisions (the reference divides every entry, esekfom.hpp:627): one more rounding
  // per entry, 1e-16 relative, in sums whose order of accumulation - MFMA blocks, workgroup partials - is not the
  // reference's either and moves them by 1e-13. -DROWS_DIVIDE builds the dividing form: measured +0.7 us on this kernel
  // (10.5 -> 11.3 us between events), no test outcome changes.
#ifndef ROWS_DIVIDE
  const double rinv = selected ? 1.0 / rc : 0.0;
#endif
  // ---- a10: the 97 sums of this workgroup as ONE 16x16 f64 outer-product accumulation on the matrix cores.
  // Per point p:  a_p = [ u/r (12) | u[0..2] (3) | 0 ],  b_p = [ u (12) | hs | 0 0 0 ];  D = sum_p a_p b_p^T holds
  // H^T R^-1 H (rows 0..11 x cols 0..11), H^T R^-1 h (col 12) and N^T N (rows 12..14 x cols 0..2).
  // v_mfma_f64_16x16x4_f64 wants A[i = lane & 15][k = lane >> 4] and B[k = lane >> 4][j = lane & 15]: the rows are
  // transposed through LDS (stride 17 doubles: conflict-free b64 writes, contiguous reads); each wave accumulates
  // its own 64 points in 16 dependent MFMAs (~0.2 us) instead of a 128-step FMA chain per entry fed by 400 KB of
  // LDS reads. The accumulation order is fixed by the hardware, so the result is still reproducible run to run.
#pragma unroll
  for (int k = 0; k < 12; k++) {
#ifndef ROWS_DIVIDE
    SA[threadIdx.x][k] = u[k] * rinv;
#else
    SA[threadIdx.x][k] = selected ? u[k] / rc : 0.0;
#endif
    SB[threadIdx.x][k] = u[k];
  }
  SA[threadIdx.x][12] = u[0], SA[threadIdx.x][13] = u[1], SA[threadIdx.x][14] = u[2], SA[threadIdx.x][15] = 0.0;
  SB[threadIdx.x][12] = hs, SB[threadIdx.x][13] = 0.0, SB[threadIdx.x][14] = 0.0, SB[threadIdx.x][15] = 0.0;
  unsigned long long bal = __ballot(selected);
  __shared__ int wcnt[BLK / 64];
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(bal);
  __syncthreads();
  PH(1, 3);
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int prow = (threadIdx.x & ~63) + (lane >> 4), col = lane & 15;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 16; g++)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(SA[prow + 4 * g][col], SB[prow + 4 * g][col], acc, 0, 0, 0);
    // C/D map of the f64 form: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int rg = 0; rg < 4; rg++) DW[wave][(lane >> 4) + 4 * rg][col] = acc[rg];
  }
  PH(1, 4);
  __syncthreads();
  // entry e: 0..77 -> (ra, cb) upper triangle of Y^T X ; 78..89 -> rhs Y^T hs ; 90..95 -> X^T X (3x3) ; 96 count
  const int e = threadIdx.x;
  if (e < NSUM) {
    double v;
    if (e == NSUM - 1) {
      v = (double)((wcnt[0] + wcnt[1]) + (wcnt[2] + wcnt[3]));
    } else {
      int ra, cb;
      if (e < 78) {
        int rem = e;
        ra = 0;
        while (rem >= 12 - ra) rem -= 12 - ra, ra++;
        cb = ra + rem;
      } else if (e < 90) {
        ra = e - 78, cb = 12;
      } else {
        const int m6[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {0, 2}, {1, 2}};
        ra = 12 + m6[e - 90][0], cb = m6[e - 90][1];
      }
      v = (DW[0][ra][cb] + DW[1][ra][cb]) + (DW[2][ra][cb] + DW[3][ra][cb]);
    }
    a.partials[(size_t)e * a.pstride + blockIdx.x] = v;
  }
  PH(1, 5);
}

#ifdef MALIO_PHASE_CLOCK
extern "C" int malio_debug_phase(long long *out64) {
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_phase), sizeof(long long) * 64) == hipSuccess ? 0 : -1;
}
extern "C" int malio_debug_span(long long *out, int n) {  // [14][n]: rows of g_span for workgroups 0..n-1
  if (n > 8192) return -1;
  for (int r = 0; r < 14; r++)
    if (hipMemcpyFromSymbol(out + (size_t)r * n, HIP_SYMBOL(g_span), sizeof(long long) * n, sizeof(long long) * 8192 * r) != hipSuccess)
      return -1;
  return 0;
}
#endif

// Fixed-order final sum, one wave per (LiDAR, entry). THE ORDER (shared by the three-kernel pass and the speculating pass,
// k_pass): a complete binary tree over the LiDAR's 64-point tiles in scan order - tile
// pairs, pairs of pairs, ... - with missing leaves read as +0 (an exact identity: no partial sum is ever -0). A workgroup
// partial of k_rows_reduce is the tree node over 4 tiles ((T0+T1)+(T2+T3)); here lane j adds four of them the same way
// ((p0+p1)+(p2+p3): two more levels) and six xor-butterfly steps with growing stride finish the node over 256 partials =
// 65 536 points (addition commutes, so both lanes of a pair hold the same bits); a LiDAR with more points than that takes
// further rounds, whose results meet in a second butterfly. All loads independent: one memory round trip deep.
// out: [L][NSUM].
struct SegBlocks {
  int b[MALIO_MAX_LIDAR + 1];
};
// (round 5: on DPP row operations like wave_max - after every step the lanes of a group hold the same bits, so any lane of the
// partner group stands for the partner of the xor step, and IEEE addition commutes: lane 63 ends with the bits of the xor
// butterfly's ((((a0 + a1) + (a2 + a3)) + ...): read back and returned to every lane. Twelve LDS-crossbar round trips less per call.)
__device__ __forceinline__ double butterfly_up(double a) {
#ifndef KS_NO_DPP
  a += dpp_f64<0xB1, 0xF>(a);   // quad_perm [1,0,3,2]
  a += dpp_f64<0x4E, 0xF>(a);   // quad_perm [2,3,0,1]
  a += dpp_f64<0x141, 0xF>(a);  // row_half_mirror
  a += dpp_f64<0x140, 0xF>(a);  // row_mirror
  a += dpp_f64<0x142, 0xA>(a);  // row_bcast:15 into rows 1 and 3 (the other rows' values are not used from here on)
  a += dpp_f64<0x143, 0xC>(a);  // row_bcast:31 into rows 2 and 3
  return readlane63_f64(a);
#else
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) a += __shfl_xor(a, sft);
  return a;
#endif
}
// LPL = leaves per lane: 4 when the leaves are k_rows_reduce's workgroup partials (each the node over 4 tiles), 16 when they
// are the tiles of k_pass themselves - the same tree either way.
// fold (k_pass' pass: nobody has folded the extrema slots yet): the first wave of workgroup 0 folds them and publishes
// [max_u, -min_u, max_R, -min_R, M, 0] next to the sums, as k_rows_reduce does on the three-kernel path.
// With a gate (gate.msg_seq set): the workgroup that finishes last - a ticket counter - announces the sums to the host
// through a sequence word in pinned memory and, in the gated update loop (gate.dl set), waits for the next pass' control
// block and installs it (gate_body): the gate costs no launch of its own.
struct FoldArgs {
  const u64 *mmslots;  // this pass' slot set (device loop: the base of both sets, parity from the control block); null: no fold
  double *mm_out;
  int extrinsic_est_en;
};
constexpr int FR16_BLK = 1024;  // workgroup of k_final_reduce<16>: four (LiDAR, entry) pairs, four waves each
template <int LPL>
__global__ void __launch_bounds__(LPL == 16 ? FR16_BLK : BLK) k_final_reduce(const double *__restrict__ partials, int pstride, SegBlocks sb,
                                                      int L, double *out, const DevLoop *dl /* device loop, or null */,
                                                      GateArgs gate /* gate.dl == null: none */, FoldArgs fold) {
  if (dl && dl->done) return;
  const int w = (int)((blockIdx.x * BLK + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (fold.mmslots && blockIdx.x == 0 && threadIdx.x < 64) {
    const u64 *slots = dl ? fold.mmslots + (size_t)dl->mm_parity * MM_SLOTS * 5 : fold.mmslots;
    double o5[5];
    mm_fold_wave(slots, fold.extrinsic_est_en, o5);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) fold.mm_out[k] = o5[k];
      fold.mm_out[5] = 0.0;
    }
  }
  if (LPL == 16) {
    // tiles of k_pass: 4 x as many leaves as workgroup partials. FOUR waves per (LiDAR, entry) - every wave takes a
    // contiguous 256 leaves of each round, lane j four of them: four loads per lane on the chain instead of sixteen -
    // whose nodes meet in LDS as (n0 + n1) + (n2 + n3): the same tree. Workgroups of 1 024 threads (four entries each), so
    // that the gate's ticket still sees ~73 arrivals (291 same-address atomics behind system-scope fences cost the pass
    // 3 us of its 0.5 us gain).
    __shared__ double s_node[FR16_BLK / 64], s_round[FR16_BLK / 256][64];
    const int grp = (int)(threadIdx.x >> 8), wv = (int)((threadIdx.x >> 6) & 3);
    const int we = (int)blockIdx.x * (FR16_BLK / 256) + grp;
    const bool live = we < L * NSUM;
    const int lid = live ? we / NSUM : 0, e = live ? we - lid * NSUM : 0;
    const int b0 = sb.b[lid], b1 = live ? sb.b[lid + 1] : sb.b[lid];
    const double *row = partials + (size_t)e * pstride;
    int rounds_max = 0;  // workgroup-uniform loop bound: the longest segment
#pragma unroll
    for (int l = 0; l < MALIO_MAX_LIDAR; l++) rounds_max = max(rounds_max, (sb.b[l + 1] - sb.b[l] + 1023) / 1024);
    const int rounds = (b1 - b0 + 1023) / 1024;
    for (int r = 0; r < rounds_max; r++) {
      const int base = b0 + 1024 * r + 256 * wv + 4 * lane;
      double p[4];
#pragma unroll
      for (int u = 0; u < 4; u++) p[u] = (r < rounds && base + u < b1) ? row[base + u] : 0.0;
      const double nd = butterfly_up((p[0] + p[1]) + (p[2] + p[3]));
      if (lane == 0) s_node[grp * 4 + wv] = nd;
      __syncthreads();
      if ((threadIdx.x & 255) == 0 && r < rounds)
        s_round[grp][r] = (s_node[grp * 4] + s_node[grp * 4 + 1]) + (s_node[grp * 4 + 2] + s_node[grp * 4 + 3]);
      __syncthreads();
    }
    if (live && wv == 0) {
      double a = lane < rounds ? s_round[grp][lane] : 0.0;
      if (rounds > 1) a = butterfly_up(a);  // (<= 64 rounds: 4 M points per LiDAR)
      if (lane == 0) out[lid * NSUM + e] = a;
    }
  } else if (w < L * NSUM) {  // wave-uniform
    const int lid = w / NSUM, e = w - lid * NSUM;
    const int b0 = sb.b[lid], b1 = sb.b[lid + 1];
    const double *row = partials + (size_t)e * pstride;
    const int rounds = (b1 - b0 + 64 * LPL - 1) / (64 * LPL);
    double vr = 0.0, a = 0.0;
    for (int r = 0; r < rounds; r++) {
      const int base = b0 + 64 * LPL * r + LPL * lane;
      double p[LPL];
#pragma unroll
      for (int u = 0; u < LPL; u++) p[u] = (base + u < b1) ? row[base + u] : 0.0;
#pragma unroll
      for (int wd = 1; wd < LPL; wd <<= 1)
#pragma unroll
        for (int u = 0; u < LPL; u += 2 * wd) p[u] = p[u] + p[u + wd];
      a = butterfly_up(p[0]);
      if (lane == r) vr = a;
    }
    if (rounds > 1) a = butterfly_up(vr);  // (<= 64 rounds: 4 M points per LiDAR)
    if (lane == 0) out[lid * NSUM + e] = a;
  }
  if (!gate.msg_seq) return;
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();  // this workgroup's sums (pinned host memory) before its ticket
    const u32 t = atomicAdd(gate.ticket, 1u);
    s_last = t == gridDim.x - 1;
    if (s_last) *gate.ticket = 0u;  // for the next kernel that carries a gate (stream-ordered)
  }
  __syncthreads();
  if (!s_last) return;
  if (gate.dl) {
    gate_body(gate);
  } else if (threadIdx.x == 0) {  // announcement only (malio_measure polls this word instead of waiting for the queue's signal)
    __threadfence_system();
    __hip_atomic_store(gate.msg_seq, gate.publish, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- the speculating pass: point phase + rows in ONE kernel (k_pass), then k_final_reduce ----------------------------------
// A pass is three kernels (k_search | k_reuse -> k_rows_reduce -> k_final_reduce) because the FIC weights of a5/a7 need the
// extrema of unit_cov and R over ALL accepted points (laserMapping.cpp:625-628,646-656,716-721): a grid-wide dependency
// between the point phase and the rows. From the second pass of a scan on those extrema are almost always the ones of
// the pass before (they belong to two or three extreme points that stay accepted), so k_pass SPECULATES on them:
//   - point phase as in k_search (search pass) or reuse_point (reuse pass), on workgroups of 64 points that never
//     straddle two LiDAR segments; the true extrema still go to the atomic slots;
//   - the control wave continues, with plane, residual and trace still in registers, into the Jacobian row (point_row,
//     weighted with the GUESSED extrema) and 16 f64 MFMAs: the 97 sums of its 64 points = one TILE, a leaf of the
//     summation tree (k_rows_reduce's workgroup partial is the node over four of them);
//   - k_final_reduce<16> adds the tiles in the tree's order behind ONE kernel boundary, folds the extrema slots and
//     announces the pass. The host compares the true extrema with the guess: equal - the sums are the pass' sums, bit for
//     bit those of the three-kernel path; different (or no guess: first pass of a scan) - the rows are redone by
//     k_rows_reduce + k_final_reduce from the per-point state k_pass left, exactly as after k_search.
// Measured and rejected (profiles/round3/r03a_one_launch_pass.txt): the whole pass as ONE launch - tiles stored
// write-through, arrival tickets, the last workgroup of every 64 adding their tiles, nodes to pinned memory, a global
// ticket for the extrema - took 50 / 38 us per search / reuse pass against 43 / 26.5 us for three kernels: three
// in-launch hand-offs between workgroups cost 9 us on the critical path, more than the boundaries they replace.
// What it buys (profiles/round3/r03b_*): the rows inside the issue-bound search kernel lengthen it by 4.9 us (23.5 ->
// 28.4 us in rocprofv3) where their own kernel takes 7.4 us, and k_final_reduce over 4 x as many leaves takes 7.4 instead
// of 5.7 us: a search pass 36.6 -> 35.8 us of kernels. A reuse pass (one active wave per workgroup here, against
// k_reuse's and k_rows_reduce's thread-per-point grids) comes out even or worse - 21.0 against 20.4 us; forming a tile per
// wave (thread = point, half-staged rows, 3 spilled VGPRs) was measured too: 16.2 against 15.6 us for the kernel - so
// malio_measure speculates on search passes only. In the enqueued-ahead update every unit is a k_pass unit: there the
// three-kernel unit pays for reading the matrix form of the state through vector loads (k_rows_reduce<true>: 82 VGPRs,
// 8.1-8.6 us) and the update gains 4-9 us.
// Not used for the dense rows of the rows path. On a map shard a workgroup of other shards' tiles stores a zero tile (malio_measure_node
// speculates on the GLOBAL extrema of the previous pass; hit or miss is decided across the shards by the exchange).
struct FuseArgs {
  int seg_start[MALIO_MAX_LIDAR + 1];
  int seg_blk0[MALIO_MAX_LIDAR + 1];  // first workgroup (64-point tile) of each LiDAR segment
  int L, converge;
  PassConst pc;     // DEV: read from the control block instead
  WeightConst wc;
  double guess[4];  // [max_u, -min_u, max_R, -min_R] the rows are weighted with (DEV: control block)
  double *tiles;    // [NSUM][tstride]: entry-major like the workgroup partials of k_rows_reduce
  int tstride;
};

__device__ __forceinline__ int tile_entry(int ra, int cb) {  // (row, col) of the 16 x 16 MFMA block -> entry of the 97, or -1
  if (ra < 12) {
    if (cb >= ra && cb < 12) return ra * 12 - (ra * (ra - 1)) / 2 + (cb - ra);
    return cb == 12 ? 78 + ra : -1;
  }
  const int a = ra - 12;
  if (a > 2 || cb > 2 || cb < a) return -1;
  if (a == cb) return 90 + a;
  return a == 0 ? (cb == 1 ? 93 : 94) : 95;
}

// (KS_QUAD) the plane weight c_i under the guessed extrema, formed by phase C' where unit_cov appears
struct CpGuess {
  static constexpr bool enabled = true;
  const WeightConst *wc;
  const double *mm;
  __device__ __forceinline__ double operator()(double ucov) const { return row_plane_weight(*wc, mm, ucov); }
};
template <bool DEV, bool SKIP>
__global__ void __launch_bounds__(KS_BLK) __attribute__((amdgpu_waves_per_eu(KS_WPE, KS_WPE)))
k_pass(Pass1Args a, NlView nl1, NlView nl2, FuseArgs f, const DevLoop *__restrict__ dl) {
  __shared__ SearchLds S;
  // the 64 rows of the control wave for the MFMA operands: u[12], hs, 1/r (a_p = [u/r | u0..2 | 0], b_p = [u | hs | 0 0 0]
  // are formed when they are read: 7.7 KB instead of the two 17-double records of k_rows_reduce)
  __shared__ double U[SQ][15];
  __shared__ RowPre RP;
  POISON_LDS(RP);
  POISON_LDS(S);
  POISON_LDS(U);
  POISON_SYNC();
  if (threadIdx.x == 0) S.nbp = reinterpret_cast<float4 (*)[SQ]>(&U[0][0]);  // (5 KB of U's 7.5: the fit is over before a row is staged)
  if (DEV && dl->done) return;
  const QuatConst &qc = DEV ? dl->qc : a.qc;
  const PassConst &pc = DEV ? dl->pc : f.pc;
  PassDyn dy;
  if (DEV) {
    const int mp = dl->mm_parity;
    dy.commit_prev = dl->commit_prev, dy.skip = dl->search_skip;
    dy.mm_cur = a.mm_base + (size_t)mp * MM_SLOTS * 5, dy.mm_next = a.mm_base + (size_t)(mp ^ 1) * MM_SLOTS * 5;
  } else {
    dy.commit_prev = a.commit_prev, dy.skip = a.skip, dy.mm_cur = a.mm_cur, dy.mm_next = a.mm_next;
  }
  const int converge = DEV ? dl->converge : f.converge;
  // this workgroup's 64 points: inside ONE LiDAR segment. `tile`: its leaf of the summation tree (position in scan order).
  int lid = 0, tile = (int)blockIdx.x;
  if (a.part.world > 1) {
    // A tile shard: the points it owns come first in every LiDAR segment (k_sort_count), and the workgroups are handed
    // out segment-interleaved - tile 0 of every LiDAR, tile 1 of every LiDAR, ... - so that ALL owned workgroups are
    // among the first the dispatcher starts. (A 200 k-point scan is 3 125 workgroups, whose launch alone takes 5 us: in
    // index order the owned workgroups of the last LiDAR entered 4.3 us after the first one's, on a kernel of 20 us -
    // profiles/round4/r04d_shard_entry.txt.) Uniform arithmetic on the kernel arguments, <= MALIO_MAX_LIDAR rounds.
    int rem = (int)blockIdx.x, base_t = 0;
    for (int round = 0; round < MALIO_MAX_LIDAR; round++) {
      int nact = 0, m = 0x7FFFFFFF;
#pragma unroll
      for (int l = 0; l < MALIO_MAX_LIDAR; l++) {
        const int left = (l < f.L ? f.seg_blk0[l + 1] - f.seg_blk0[l] : 0) - base_t;
        if (left > 0) nact++, m = min(m, left);
      }
      if (nact == 0) break;
      if (rem < m * nact) {
        const int which = rem % nact;
        int seen = 0;
#pragma unroll
        for (int l = 0; l < MALIO_MAX_LIDAR; l++) {
          const int left = (l < f.L ? f.seg_blk0[l + 1] - f.seg_blk0[l] : 0) - base_t;
          if (left > 0 && seen++ == which) lid = l;
        }
        tile = f.seg_blk0[lid] + base_t + rem / nact;
        break;
      }
      rem -= m * nact, base_t += m;
    }
  } else {
    // Workgroup -> tile, XCD-aware. Consecutive workgroup ids go round-robin over the 8 XCDs, each with an L2 of its own, and a
    // tile's 97 sums are 97 eight-byte stores into 97 different lines of the entry-major tile array: with tile == id the 16 tiles
    // that share a 128-byte line came from all 8 XCDs and every L2 wrote its pieces back as partial lines at the end of the
    // kernel (WRITE_SIZE 5.1 MB for 1.2 MB of tiles, 2.0 us of k_pass: profiles/round5/r05b). Inside every block of 128 ids XCD x
    // forms the 16 CONSECUTIVE tiles [16 x, 16 x + 16): -1.0 us. (Measured and not used, profiles/round5/r05d, r05i, r05q, r05s: one
    // contiguous eighth of the scan per XCD, the x-th eighth of every LiDAR segment per XCD - a third fewer fetched bytes, not a
    // microsecond -, tiles dealt to the CUs by weight, a rotated sub-block per 128-block.)
    {
      const int b = (int)blockIdx.x, full = (int)gridDim.x & ~127;
      if (b < full) tile = (b & ~127) + ((b & 7) << 4) + ((b & 127) >> 3);
      // ... in REVERSE scan order: the launch hands out ~1 600 workgroups over ~1 us, lowest id first, and the workgroups
      // that finish last are those of the sparser LiDARs at the END of the scan (fewer queries share a cell: more distinct
      // lines to fetch per workgroup, profiles/round5/r05h) - they enter first. The ids past the last whole block of 128 are handed
      // out LAST and land as a SEVENTH workgroup on CUs that hold six (1 564 workgroups over 256 CUs at BASELINE config 2): these
      // were the kernel's last workgroups to leave - by 2 us, whatever the rest gained (profiles/round5/r05p_*). They take the
      // scan's FIRST tiles (the densest LiDAR: the lightest), everything else moves up.
      tile = b < full ? (int)gridDim.x - 1 - tile : b - full;
    }
#pragma unroll
    for (int l = 1; l < MALIO_MAX_LIDAR; l++)
      if (l < f.L && tile >= f.seg_blk0[l]) lid = l;
  }
  lid = __builtin_amdgcn_readfirstlane(lid), tile = __builtin_amdgcn_readfirstlane(tile);  // (workgroup-uniform: scalars)
  const int q0 = f.seg_start[lid] + (tile - f.seg_blk0[lid]) * SQ, qend = f.seg_start[lid + 1];
  const int lane = (int)(threadIdx.x & 63);
  PointOut po;
  int role = ROLE_CONTROL;
#if KS_QUAD
  double mm[4];  // the guessed extrema the rows are weighted with
#pragma unroll
  for (int k = 0; k < 4; k++) mm[k] = DEV ? dl->mm_guess[k] : f.guess[k];
#endif
  if (converge) {
#if KS_QUAD
    // (phase A's shadow, helper wave, lane = query: both traces, point_this and 1 / R_i - the row's factors that need the scan
    // point and the guess, not the search)
    auto pre_a = [&](int ln, int i, bool in) {
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      double trS = 0.0, trR = 0.0;
      if (in) {
        q = a.scan[i];
        const int packed = __float_as_int(q.w);
        trace_both(a, q, packed & 0xFF, packed >> 8, trS, trR);
      }
      S.trS[ln] = trS, S.trR[ln] = trR;
      const D3 X = row_point_imu(pc, lid, D3{(double)q.x, (double)q.y, (double)q.z});
      RP.X[0][ln] = X.x, RP.X[1][ln] = X.y, RP.X[2][ln] = X.z;
      double rc = row_point_noise(f.wc, a.extrinsic_est_en, mm, trS);
      if (rc < 0.0001) rc = 0.001;  // esekfom.hpp:624-626
#ifndef ROWS_DIVIDE
      RP.rw[ln] = 1.0 / rc;
#else
      RP.rw[ln] = rc;
#endif
    };
    role = search_wg<DEV, SKIP, KS_PIPE != 0>(a, nl1, nl2, qc, dy, S, q0, qend, po, pre_a, CpGuess{&f.wc, mm}, RP.cp);
#else
    role = search_wg<DEV, SKIP, KS_PIPE != 0>(a, nl1, nl2, qc, dy, S, q0, qend, po);
#endif
    if (role == ROLE_RETIRE) return;  // (the search waves retire; the second wave stays as the helper)
    if (po.skipped) {  // a workgroup of another shard's tiles: its leaf of the summation tree is a zero tile, nothing else
      for (int e = lane; e < NSUM; e += 64) f.tiles[(size_t)e * f.tstride + tile] = 0.0;
      return;
    }
  } else {  // REUSE pass: the control wave (and its helper), lane = point
    if (blockIdx.x == 0 && threadIdx.x < MM_SLOTS) mm_reset_slot(dy.mm_next, threadIdx.x);
    if (threadIdx.x >= 128) return;
    role = threadIdx.x >= 64 ? ROLE_HELPER : ROLE_CONTROL;
    const int i = q0 + lane;
    po.selected = false, po.ucov = 0.0, po.tr = 0.0, po.pd2 = 0.f;
    po.pl = make_float4(0.f, 0.f, 0.f, 0.f), po.q = po.pl;
    if (a.part.world > 1 && !__ballot(i < qend && a.nfound[i] != NF_NOTMINE)) {  // a workgroup of other shards' points: a zero tile
      if (role == ROLE_CONTROL)
        for (int e = lane; e < NSUM; e += 64) f.tiles[(size_t)e * f.tstride + tile] = 0.0;
      return;  // (both waves see the same ballot)
    }
    if (role == ROLE_CONTROL) {
      float4 wld = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < qend) reuse_point_ctrl(a, qc, i, po.selected, po.pl, po.pd2, po.q, wld);
      S.w[lane] = wld;
    }
  }
  PH(2, 0);
  // ---- a5 / a7 with the guessed extrema ----
#if !KS_QUAD
  double mm[4];
#pragma unroll
  for (int k = 0; k < 4; k++) mm[k] = DEV ? dl->mm_guess[k] : f.guess[k];
#endif
#if KS_QUAD
  if (converge && role == ROLE_HELPER) {  // (phase C' left flag, plane, residual, unit_cov and c_i in LDS; its barrier is behind us)
    const int i = q0 + lane;
    helper_post<SKIP>(a, dy.mm_cur, S, lane, i, i < qend && S.w[lane].x < 1e9f, S.nf[lane]);  // per-point state; a4: the TRUE extrema
    return;
  }
#endif
  if (role == ROLE_HELPER) {  // (see helper_unit_cov_trace)
    const int i = q0 + lane;
    bool served;
    int nf;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    double ucov = 0.0, trS = 0.0;
    unsigned char sel_old = 0;  // (reuse pass: what the lazy normal_y fold of the previous pass reads, commit_normal_y)
    double trace_old = 0.0;
    PHH(3, 0);
    if (converge) {
      served = i < qend && S.w[lane].x < 1e9f;
      nf = S.nf[lane];
      u32 og[5];
#pragma unroll
      for (int k = 0; k < 5; k++) og[k] = S.og[k][lane];
      q = S.q[lane];
      helper_unit_cov_trace(a, S, lane, i, served, nf, og, q, ucov, trS);
    } else {  // the plane and its unit_cov are the search pass': a point accepted now has nf == 5 and a stored unit_cov
      nf = i < qend ? (int)a.nfound[i] : 0;
      served = i < qend && nf != NF_NOTMINE;
      double trR = 0.0;
      if (served) {
        q = a.scan[i];
        if (nf == 5) ucov = a.ucov[i];
        if (dy.commit_prev) sel_old = a.sel[i], trace_old = a.trace[i];
        const int packed = __float_as_int(q.w);
        trace_both(a, q, packed & 0xFF, packed >> 8, trS, trR);
      }
      S.ucv[lane] = ucov, S.trS[lane] = trS, S.trR[lane] = trR;
    }
    PHH(3, 1);
    // the plane-independent factors of the row, for the case the point is accepted
    const D3 X = row_point_imu(pc, lid, D3{(double)q.x, (double)q.y, (double)q.z});
    RP.X[0][lane] = X.x, RP.X[1][lane] = X.y, RP.X[2][lane] = X.z;
    RP.cp[lane] = row_plane_weight(f.wc, mm, ucov);
    double rc = row_point_noise(f.wc, a.extrinsic_est_en, mm, trS);
    if (rc < 0.0001) rc = 0.001;  // esekfom.hpp:624-626
#ifndef ROWS_DIVIDE
    RP.rw[lane] = 1.0 / rc;
#else
    RP.rw[lane] = rc;
#endif
    PHH(3, 2);
    __syncthreads();
    PHH(3, 3);
    if (converge)
      helper_post<SKIP>(a, dy.mm_cur, S, lane, i, served, nf);  // per-point state; a4: the TRUE extrema of this pass
    else
      helper_post_reuse(a, dy.mm_cur, S, lane, i, served, dy.commit_prev, sel_old, trace_old);
    PHH(3, 4);
    return;
  }
  if (!(KS_QUAD && converge)) {  // (a search pass under KS_QUAD: phase C' has handed everything over already)
    S.selc[lane] = po.selected ? 1 : 0, S.plc[lane] = po.pl, S.pd2c[lane] = po.pd2;
    PH(2, 6);
    __syncthreads();
  }
  PH(2, 1);
  double u[12], hs = 0;
#pragma unroll
  for (int k = 0; k < 12; k++) u[k] = 0;
  if (po.selected)
    row_finish(a.extrinsic_est_en, pc, lid, po.q, po.pl, po.pd2, D3{RP.X[0][lane], RP.X[1][lane], RP.X[2][lane]}, RP.cp[lane], u, hs);
  PH(2, 2);
#pragma unroll
  for (int k = 0; k < 12; k++) U[lane][k] = u[k];
  U[lane][12] = hs;
  U[lane][13] = po.selected ? RP.rw[lane] : 0.0;
  const unsigned long long bal = __ballot(po.selected);
  __builtin_amdgcn_wave_barrier();  // one wave: its LDS stores above precede its loads below (waitcnt by the compiler)
  PH(2, 3);
  const int prow = lane >> 4, col = lane & 15;
  const int ca = col < 12 ? col : (col < 15 ? col - 12 : 0), cb = col < 12 ? col : 12;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  // Operands of the 16 dependent MFMAs, read eight iterations ahead and formed WITHOUT branches: a lane-dependent
  // `col < 12 ? x * w : ...` around the LDS read of w compiled into an exec-masked branch and an s_waitcnt lgkmcnt(0) in
  // front of every MFMA - 1.3 us for this loop (phase clocks, profiles/round4) against ~0.3 us of MFMA latency. The
  // selects below pick the same values: x * 1.0 == x bit for bit, and row 15 / columns 13-15 of the block are not read.
#ifndef ROWS_DIVIDE
  const double wsel = col < 12 ? 0.0 : 1.0;  // added to nothing: selects w or 1.0 below
#endif
#pragma unroll
  for (int half = 0; half < 2; half++) {
    double xa[8], wa[8], ya[8];
#pragma unroll
    for (int g = 0; g < 8; g++) {
      const double *up = U[prow + 4 * (8 * half + g)];
      xa[g] = up[ca], wa[g] = up[13], ya[g] = up[cb];
    }
#pragma unroll
    for (int g = 0; g < 8; g++) {
      const double x = col < 15 ? xa[g] : 0.0, y = col <= 12 ? ya[g] : 0.0;
#ifndef ROWS_DIVIDE
      const double av = x * (col < 12 ? wa[g] : wsel);
#else
      const double av = col < 12 ? (wa[g] != 0.0 ? x / wa[g] : 0.0) : x;
#endif
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, y, acc, 0, 0, 0);
    }
  }
  PH(2, 4);
  // ---- the tile: a leaf of the summation tree, entry-major like k_rows_reduce's partials ----
#ifndef ATTR_NO_TILES
#pragma unroll
  for (int rg = 0; rg < 4; rg++) {
    const int e = tile_entry(prow + 4 * rg, col);
    if (e >= 0) f.tiles[(size_t)e * f.tstride + tile] = acc[rg];
  }
  if (lane == 0) f.tiles[(size_t)(NSUM - 1) * f.tstride + tile] = (double)__popcll(bal);
#else
  if (acc[0] == 1.2345e300 && lane == 0) f.tiles[tile] = acc[1] + acc[2] + acc[3] + (double)__popcll(bal);  // (keeps the MFMAs alive)
#endif
  PH(2, 5);
  PH_EXIT();
}


// ---- the REUSE pass shaped for streaming (round 6): k_reuse_rows -> k_final_reduce<4> ------------------------------------------
// A reuse pass (ekfom_data.converge == false, laserMapping.cpp:583-595) keeps neighbours and plane: 36 bytes per point in, a few
// out, no search - a streaming job. As k_reuse -> k_rows_reduce -> k_final_reduce it was three launches (26.8 us for 3.6 MB at
// BASELINE config 2), as k_pass' reuse form one wave of every 64-point workgroup. Here: thread = point over ALL four waves of a
// 256-point workgroup of ONE LiDAR (k_rows_reduce's blocks) - world point, cached plane, gates, trace (reuse_point), the extrema
// to the atomic slots - and, speculating on the previous pass' extrema like k_pass (section 3.1 of DESIGN.md), the a5/a7 row
// straight from the registers, staged as [u | hs | 1/r] (k_pass' 15-double record: 30 KB), one MFMA tile per wave, the four
// tiles added (T0 + T1) + (T2 + T3): THE SAME workgroup partial, bit for bit, k_rows_reduce forms from the per-point state
// (same expressions on the same operands: point_row; same 16 dependent MFMAs; same tree) - so a wrong guess is repaired by
// k_rows_reduce + k_final_reduce<4> exactly as after k_pass, and the summation tree never notices which kernel ran.
// k_final_reduce<4> folds the extrema slots (FoldArgs) and carries the gate. Used by malio_measure, the host-driven loop and
// malio_measure_node for every reuse pass that may speculate; the enqueued-ahead chain keeps k_pass<true, .> (a unit does not
// know in advance which kind of pass it will be).
struct ReuseRowsArgs {
  int seg_block0[MALIO_MAX_LIDAR + 1];  // first 256-point workgroup of each LiDAR segment
  int seg_start[MALIO_MAX_LIDAR + 1];
  int L;
  PassConst pc;
  WeightConst wc;
  double guess[4];   // [max_u, -min_u, max_R, -min_R] the rows are weighted with
  double *partials;  // [NSUM][pstride], entry-major: k_rows_reduce's
  int pstride;
};
__global__ void __launch_bounds__(BLK) k_reuse_rows(Pass1Args a, ReuseRowsArgs f) {
  __shared__ double U[BLK][15];
  __shared__ double DW[BLK / 64][16][16];
  __shared__ int wcnt[BLK / 64];
  POISON_LDS(U);
  POISON_LDS(DW);
  POISON_SYNC();
  const PassDyn dy = pass_dyn<false>(a);
  int lid = 0;
#pragma unroll
  for (int l = 1; l < MALIO_MAX_LIDAR; l++)
    if (l < f.L && (int)blockIdx.x >= f.seg_block0[l]) lid = l;
  const int i = f.seg_start[lid] + ((int)blockIdx.x - f.seg_block0[lid]) * BLK + (int)threadIdx.x;
  const bool in = i < f.seg_start[lid + 1];
  // reuse_point's arithmetic and stores with ALL of the point's loads in one round trip (behind its flags they were four
  // dependent ones: this kernel is as long as its chain of trips, 14.6 us at BASELINE config 2 AND at config 1's tenth of the
  // points), and the trace under both clamp rules as soon as the scan point is there (trace_both: the table entry's load does
  // not wait for the gate) - the flag picks one, bit for bit trace_for's.
  bool selected = false;
  double ucov = 0.0, tr = 0.0;
  float4 pl = make_float4(0.f, 0.f, 0.f, 0.f), q = pl;
  float pd2 = 0.f;
  {
    unsigned char nfo = NF_NOTMINE, sel_old = 0;
    double ucov_st = 0.0, trace_old = 0.0;
    float4 pl_st = pl;
    if (in) {
      q = a.scan[i], nfo = a.nfound[i], sel_old = a.sel[i], pl_st = a.plane[i], ucov_st = a.ucov[i];
      if (dy.commit_prev) trace_old = a.trace[i];
    }
    if (in && nfo != NF_NOTMINE) {  // (a partitioned handle keeps serving the points of its last search pass)
      const int packed = __float_as_int(q.w);
      const int lid_p = packed & 0xFF;
      double trS, trR;
      trace_both(a, q, lid_p, packed >> 8, trS, trR);
      float wx, wy, wz;
      double nb;
      world_point(a.qc, q, lid_p, wx, wy, wz, nb);
      a.world[i] = wx, a.world[a.N + i] = wy, a.world[2 * a.N + i] = wz;
      if (dy.commit_prev && !(sel_old && !a.extrinsic_est_en)) a.ny[i] = (float)trace_old;  // commit_normal_y
      if (sel_old) {
        pl = pl_st, ucov = ucov_st;
        const float pabcd[4] = {pl.x, pl.y, pl.z, pl.w};
        float p2;
        if (residual_gate(pabcd, wx, wy, wz, sqrt(nb), p2)) selected = true, a.pd2[i] = p2, pd2 = p2;
      }
      a.sel[i] = selected ? 1 : 0;
      tr = selected ? trS : trR;
      a.trace[i] = tr;
    } else {
      q = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  wave_minmax_publish(a, dy.mm_cur, selected, ucov, tr);  // a4: the TRUE extrema of this pass
  if (blockIdx.x == 0 && threadIdx.x < MM_SLOTS) mm_reset_slot(dy.mm_next, threadIdx.x);
  double u[12], hs = 0, r = 1;
#pragma unroll
  for (int k = 0; k < 12; k++) u[k] = 0;
  if (selected) {
    RowIn rin;
    rin.q = q, rin.pl = pl, rin.ucov = ucov, rin.trace = tr, rin.pd2 = pd2;
    point_row(f.wc, a.extrinsic_est_en, f.pc, f.guess, rin, lid, u, hs, r);
  }
  double rc = r;
  if (rc < 0.0001) rc = 0.001;  // esekfom.hpp:624-626
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  double *mine = U[threadIdx.x];
#pragma unroll
  for (int k = 0; k < 12; k++) mine[k] = u[k];
  mine[12] = hs;
#ifndef ROWS_DIVIDE
  mine[13] = selected ? 1.0 / rc : 0.0;
#else
  mine[13] = selected ? rc : 0.0;
#endif
  const unsigned long long bal = __ballot(selected);
  if (lane == 0) wcnt[wave] = __popcll(bal);
  __builtin_amdgcn_wave_barrier();  // (a wave reads only the 64 records it wrote)
  {
    // the tile of this wave's 64 points: k_pass' operand forms (x * 1.0 == x, selects instead of branches), k_rows_reduce's bits
    const int prow = wave * 64 + (lane >> 4), col = lane & 15;
    const int ca = col < 12 ? col : (col < 15 ? col - 12 : 0), cb = col < 12 ? col : 12;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#ifndef ROWS_DIVIDE
    const double wsel = col < 12 ? 0.0 : 1.0;
#endif
#pragma unroll
    for (int half = 0; half < 2; half++) {
      double xa[8], wa[8], ya[8];
#pragma unroll
      for (int g = 0; g < 8; g++) {
        const double *up = U[prow + 4 * (8 * half + g)];
        xa[g] = up[ca], wa[g] = up[13], ya[g] = up[cb];
      }
#pragma unroll
      for (int g = 0; g < 8; g++) {
        const double x = col < 15 ? xa[g] : 0.0, y = col <= 12 ? ya[g] : 0.0;
#ifndef ROWS_DIVIDE
        const double av = x * (col < 12 ? wa[g] : wsel);
#else
        const double av = col < 12 ? (wa[g] != 0.0 ? x / wa[g] : 0.0) : x;
#endif
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, y, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int rg = 0; rg < 4; rg++) DW[wave][(lane >> 4) + 4 * rg][col] = acc[rg];
  }
  __syncthreads();
  const int e = (int)threadIdx.x;
  if (e < NSUM) {
    double v;
    if (e == NSUM - 1) {
      v = (double)((wcnt[0] + wcnt[1]) + (wcnt[2] + wcnt[3]));
    } else {
      int ra, cb;
      if (e < 78) {
        int rem = e;
        ra = 0;
        while (rem >= 12 - ra) rem -= 12 - ra, ra++;
        cb = ra + rem;
      } else if (e < 90) {
        ra = e - 78, cb = 12;
      } else {
        const int m6[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {0, 2}, {1, 2}};
        ra = 12 + m6[e - 90][0], cb = m6[e - 90][1];
      }
      v = (DW[0][ra][cb] + DW[1][ra][cb]) + (DW[2][ra][cb] + DW[3][ra][cb]);
    }
    f.partials[(size_t)e * f.pstride + blockIdx.x] = v;
  }
}

// ---- batched Nearest_Search -----------------------------------------------------------------------
constexpr int NL2_G = 16;  // lanes per query on the level-2 lists (~180 candidates), batched API

// Batched Nearest_Search API: level-2 lists, radius limit just under cf2 (what the block always guarantees).
__global__ void __launch_bounds__(BLK) k_nearest(const float4 *__restrict__ q, int n, int k, NlView nl, u32 *out_idx,
                                                 float *out_d2, int *out_cnt) {
  const int tid = blockIdx.x * BLK + threadIdx.x;
  const int qi = tid / NL2_G, sub = tid % NL2_G;
  const bool active = qi < n;
  float4 p = q[active ? qi : n - 1];
  Top5 t;
  float lb2_unused;
  nl_search<NL2_G, false, KS_PIPE != 0>(nl, p.x, p.y, p.z, sub, nl.cf * nl.cf * 0.999f, t, lb2_unused);
  if (!active || sub != 0) return;
  int c = 0;
  for (int j = 0; j < 5; j++) {
    if (j < k) {
      bool ok = t.og(j) != INVALID;
      out_idx[(size_t)qi * k + j] = t.og(j);
      out_d2[(size_t)qi * k + j] = ok ? t.d(j) : INFINITY;
      c += ok;
    }
  }
  out_cnt[qi] = c;
}

// early: ending a walk early saves LINES - what a full GPU's level-1 search queues for - and costs the
// unsettled queries a dependent trip: on a scan that leaves most CUs with one workgroup or none (BASELINE config 1: 157
// workgroups) there is no queue to shorten and the trip is all there is (+1.2 us per pass, measured) - such a scan walks its
// lists whole. The threshold is the handle's (MALIO_OPT_EARLY_MIN_QUERIES, default 32 768 = 512 workgroups: two per CU; 0: every
// scan cuts - what the edge-case tests run); a tile shard serves 1 / world of its scan's points: that is its number of queries.
static NlView view_of(const NList &nl, int early = 0) {
  NlView v;
  v.table = nl.table, v.tmask = nl.tmask, v.pts = nl.pts, v.cf = nl.cf, v.inv_cf = nl.inv_cf;
  v.reach_cf = nl.pruned ? NL_REACH * nl.cf : nl.cf;
  v.early = early;
  return v;
}
static NlView view_l1(const Ctx *c) {
  const int nq = c->part.world > 1 ? c->N / c->part.world : c->N;
  return view_of(c->nl1, nq >= c->opt_early_min_queries ? 1 : 0);
}

int nearest_search(Ctx *c, const float4 *d_q, int n, int k, u32 *d_idx, float *d_d2, int *d_cnt) {
  if (int rc = map_sync_search(c)) return rc;
  long long threads = (long long)n * NL2_G;
  hipLaunchKernelGGL(k_nearest, dim3((unsigned)((threads + BLK - 1) / BLK)), dim3(BLK), 0, c->stream, d_q, n, k,
                     view_of(c->nl2), d_idx, d_d2, d_cnt);
  MALIO_HIP(hipGetLastError());
  return MALIO_OK;
}

// ---- map_incremental() selection (laserMapping.cpp:398-442) ------------------------------------------------
// The reference's Nearest_Points[i] holds the 5 nearest map points at ANY distance (ikd_Tree Nearest_Search has no
// radius), the search pass only kept those with d2 <= 5. That is enough for the re-add test below (a neighbour
// farther than sqrt(5) m can never be closer to the voxel centre than the point itself, which sits inside its
// 0.5 m voxel), but :421-425 also looks at points_near[0] when it is far away. k_far_nearest finds that one
// exactly for the queries whose search ball was empty: shells of 3x3x3-cell blocks of the level-2 lists around
// the query cell, until the best distance is inside the covered cube; then (rare) a scan of the whole map.
constexpr int FAR_RMAX = 6;
constexpr int FAR_GROUP = 16;  // scan points per wave and stride of k_far_nearest (a map frontier full of empty balls still fills the GPU)
// K = 1: the nearest map point of the queries with an empty search ball (what :421-425 reads), far_idx[N].
// K = 5: the unrestricted 5-NN of every query with fewer than five neighbours inside sqrt(5) m, far_idx[5][N] - what
// ikdtree.Nearest_Search leaves in Nearest_Points[i] (ikd_Tree.cpp:426-461 has no radius), handed out by malio_scan_get.
template <int K>
__global__ void __launch_bounds__(BLK) k_far_nearest(int N, const float4 *__restrict__ world4,
                                                     const unsigned char *__restrict__ nfound, NlView nl,
                                                     const float4 *__restrict__ map_in, int map_n, u32 *far_idx) {
  // A grid of waves strides over the scan FAR_GROUP points at a time: one coalesced look at the flags, INVALID for the
  // points that need nothing (nearly all of them), then the few that do are served one after the other by the whole wave.
  // (One wave per scan point - 25 k workgroups that return at once - cost 10 us for a handful of searches.)
  const int lane = threadIdx.x & 63;
  const int wave0 = (int)((blockIdx.x * BLK + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * BLK) >> 6);
  for (int base = wave0 * FAR_GROUP; base < N; base += nwaves * FAR_GROUP) {
  const int ql = base + lane;
  const bool mine = lane < FAR_GROUP && ql < N;
  // (NF_NOTMINE counts as served: another shard answers for that point)
  const bool needy = mine && (K == 1 ? nfound[ql] == 0 : nfound[ql] < 5);
  if (mine && !needy) {
#pragma unroll
    for (int k = 0; k < K; k++) far_idx[(size_t)k * N + ql] = INVALID;
  }
  unsigned long long todo = __ballot(needy);
  while (todo) {
  const int qi = base + __ffsll((long long)todo) - 1;
  todo &= todo - 1;
  const float4 w = world4[qi];
  const float gx = w.x * nl.inv_cf, gy = w.y * nl.inv_cf, gz = w.z * nl.inv_cf;
  const int cx = (int)floorf(gx), cy = (int)floorf(gy), cz = (int)floorf(gz);
  const float margin = 6e-7f * (fabsf(gx) + fabsf(gy) + fabsf(gz) + 3.0f) * nl.cf;
  // lane-local candidates under the total order (d2, map index); a deleted slot (x = +inf) is at infinite distance
  Top5 t;
#pragma unroll
  for (int k = 0; k < 5; k++) t.k[k] = TOP5_MAXKEY;
  auto offer = [&](const float4 p, u32 og) {
    float ddx = w.x - p.x, ddy = w.y - p.y, ddz = w.z - p.z;
    float d2 = ddx * ddx + ddy * ddy + ddz * ddz;  // calc_dist, ikd_Tree.cpp:1697 (no FMA)
    top5_insert(t, d2 < INFINITY ? top5_key(d2, og) : TOP5_MAXKEY);
  };
  bool done = false;
  for (int r = 0; r <= FAR_RMAX && !done; r++) {
    const int side = 2 * r + 1, nblk = side * side * side;
    for (int base = 0; base < nblk; base += 64) {
      const int bi = base + lane;
      u32 start = 0, count = 0;
      if (bi < nblk) {
        const int a = bi % side - r, b = (bi / side) % side - r, c = bi / (side * side) - r;
        if (max(max(abs(a), abs(b)), abs(c)) == r) {  // only the new shell (the blocks are disjoint: no duplicates)
          u64 key = cell_key_d(cx + 3 * a, cy + 3 * b, cz + 3 * c);
          u32 slot = hash_key_d(key) & nl.tmask;
          cell_lookup(nl.table, nl.tmask, key, nl.table[slot], slot, start, count);
          count &= NL_COUNT;
        }
      }
      unsigned long long m = __ballot(count > 0);
      while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const u32 s0 = __shfl(start, src), cn = __shfl(count, src);
        for (u32 j = (u32)lane; j < cn; j += 64) {
          const float4 p = nl.pts[(size_t)s0 + j];
          offer(p, __float_as_uint(p.w));
        }
      }
    }
    // best K over the wave; the merged list then lives in lane 0 only, so that a further shell cannot count it twice
    u32 ev_unused = 0xFFFFFFFFu;
    merge_group<64, false>(t, ev_unused);
    // everything within (3r+1) cell edges of the query's cell has been seen
    const float reach = (float)(3 * r + 1) * nl.cf - margin;
    if (t.og(K - 1) != INVALID && t.d(K - 1) <= reach * reach * 0.99999f) done = true;
    if (!done && lane != 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) t.k[k] = TOP5_MAXKEY;
    }
  }
  if (!done) {  // farther than ~40 m from every map point (or fewer than K points in the map): scan the map
#pragma unroll
    for (int k = 0; k < 5; k++) t.k[k] = TOP5_MAXKEY;
    for (int j = lane; j < map_n; j += 64) offer(map_in[j], (u32)j);
    u32 ev_unused = 0xFFFFFFFFu;
    merge_group<64, false>(t, ev_unused);
  }
  if (lane < K) far_idx[(size_t)lane * N + qi] = lane == 0 ? t.og(0) : lane == 1 ? t.og(1) : lane == 2 ? t.og(2) : lane == 3 ? t.og(3) : t.og(4);
  }  // needy queries of this group of 64
  }  // groups of 64 points
}

struct MapIncArgs {
  int N, map_n, flg_EKF_inited, extrinsic_est_en, commit_prev;
  const float4 *scan;  // sorted
  const u32 *perm;     // sorted -> original
  QuatConst qc;        // state_point (posterior) + temporal compensation
  const unsigned char *nfound, *sel;
  const u32 *nbr, *far_idx;
  const float4 *map_in;
  const float *ny;
  const double *trace;
  const float *wny;  // [N] original order: feats_down_world[i].normal_y as the caller holds it
  double cov_threshold, fs;
  u32 *addf, *nonf;  // [N + 1] original order: PointToAdd / PointNoNeedDownsample membership
  float4 *wp;        // [N] original order: the world point that would be pushed
};

__global__ void __launch_bounds__(BLK) k_mapinc_classify(MapIncArgs a) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= a.N) return;
  if (i == 0) a.addf[a.N] = 0u, a.nonf[a.N] = 0u;  // the closing zero the exclusive scans of the flags expect
  const u32 o = a.perm[i];
  const float4 q = a.scan[i];
  const int lid = (int)(__float_as_uint(q.w) & 0xFF);
  // feats_down_body[i].normal_y with the last pass' rewrite folded in (see commit_normal_y)
  if (a.nfound[i] == NF_NOTMINE) {  // another shard classifies this point
    a.addf[o] = 0u, a.nonf[o] = 0u, a.wp[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const bool untouched = !a.commit_prev || (a.sel[i] && !a.extrinsic_est_en);
  const float nyv = untouched ? a.ny[i] : (float)a.trace[i];
  u32 cls = 0;
  float wx = 0.f, wy = 0.f, wz = 0.f;
  if (!((double)nyv > a.cov_threshold)) {  // :406
    // pointBodyToWorld(PointType*, PointType*), :134-147
    const D3 p{(double)q.x, (double)q.y, (double)q.z};
    const D3 X = (lid == 0) ? qrot(a.qc.ql[0], p) + a.qc.tl[0]
                            : qrot(a.qc.qtc[lid], qrot(a.qc.ql[lid], p) + a.qc.tl[lid]) + a.qc.ttc[lid];
    const D3 pg = qrot(a.qc.rot, X) + a.qc.pos;
    wx = (float)pg.x, wy = (float)pg.y, wz = (float)pg.z;
    cls = 1;
    if (a.map_n > 0 && a.flg_EKF_inited) {  // :411 (Nearest_Search on a non-empty tree never returns nothing)
      const float mx = (float)(floor((double)wx / a.fs) * a.fs + 0.5 * a.fs);
      const float my = (float)(floor((double)wy / a.fs) * a.fs + 0.5 * a.fs);
      const float mz = (float)(floor((double)wz / a.fs) * a.fs + 0.5 * a.fs);
      const float dist = ((wx - mx) * (wx - mx) + (wy - my) * (wy - my)) + (wz - mz) * (wz - mz);
      int nf = a.nfound[i];
      if (nf > 5) nf = 0;  // defence: only 0..5 are ever stored (a search marker must never index nbr[5][N])
      const u32 n0 = nf > 0 ? a.nbr[i] : a.far_idx[i];
      const float4 m0 = a.map_in[n0];
      if ((double)fabsf(m0.x - mx) > 0.5 * a.fs && (double)fabsf(m0.y - my) > 0.5 * a.fs &&
          (double)fabsf(m0.z - mz) > 0.5 * a.fs) {  // :421-425
        cls = 2;
      } else if (a.map_n >= 5) {  // :426-435 (points_near.size() == 5 whenever the tree holds 5 points)
        for (int k = 0; k < nf; k++) {
          const float4 m = a.map_in[a.nbr[(size_t)k * a.N + i]];
          const float d = ((m.x - mx) * (m.x - mx) + (m.y - my) * (m.y - my)) + (m.z - mz) * (m.z - mz);
          if (d < dist) {
            cls = 0;
            break;
          }
        }
      }
    }
  }
  a.addf[o] = cls == 1 ? 1u : 0u;
  a.nonf[o] = cls == 2 ? 1u : 0u;
  a.wp[o] = make_float4(wx, wy, wz, a.wny ? a.wny[o] : 0.f);
}

// Nearest_Points beyond the search radius (malio_scan_get): d_far [5][N], INVALID where the search pass found all five
int far_knn5(Ctx *c, u32 *d_far) {
  const long long th = ((long long)c->N + FAR_GROUP - 1) / FAR_GROUP * 64;
  hipLaunchKernelGGL(k_far_nearest<5>, dim3((unsigned)std::min<long long>((th + BLK - 1) / BLK, 2048)), dim3(BLK), 0, c->stream, c->N, c->d_world4,
                     c->d_nfound, view_of(c->nl2), c->d_map_in, c->map_n, d_far);
  MALIO_HIP(hipGetLastError());
  return MALIO_OK;
}

int mapinc_classify(Ctx *c, const malio_state_t *state_point, int flg_EKF_inited, const float *d_wny, u32 *d_addf,
                    u32 *d_nonf, float4 *d_wp) {
  if (c->N <= 0 || !c->scan_sorted) return MALIO_ERR_NO_SCAN;
  if (int rc = map_sync_search(c)) return rc;  // (a rebuild renumbers the map: caught by the epoch test below)
  if (c->map_n > 0 && c->nbr_epoch != c->map_epoch) {
    c->err = "map_incremental: the map changed after the last search pass of this scan";
    return MALIO_ERR_BAD_ARG;
  }
  // The re-add test of :426-435 looks at the neighbours the search pass kept (d2 <= 5). That equals the reference's
  // unrestricted 5-NN only while a neighbour farther than sqrt(5) m cannot be closer to the voxel centre than the point
  // itself, i.e. while the half diagonal of a voxel stays below sqrt(5)/2: filter_size_map < sqrt(5/3) m.
  if (c->prm.filter_size_map >= 1.29) {
    c->err = "map_incremental: filter_size_map >= 1.29 m is not supported (the cached neighbours inside sqrt(5) m no longer decide laserMapping.cpp:426-435)";
    return MALIO_ERR_BAD_ARG;
  }
  const int N = c->N;
  ArenaScope sc(c->arena);
  u32 *d_far = nullptr;
  MALIO_HIP(sc.get(&d_far, (size_t)N));
  if (c->map_n - c->map_dead > 0 && flg_EKF_inited) {
    long long th = ((long long)N + FAR_GROUP - 1) / FAR_GROUP * 64;
    hipLaunchKernelGGL(k_far_nearest<1>, dim3((unsigned)std::min<long long>((th + BLK - 1) / BLK, 2048)), dim3(BLK), 0, c->stream, N, c->d_world4,
                       c->d_nfound, view_of(c->nl2), c->d_map_in, c->map_n, d_far);
  }
  MapIncArgs a;
  a.N = N, a.map_n = c->map_n - c->map_dead, a.flg_EKF_inited = flg_EKF_inited, a.extrinsic_est_en = c->prm.extrinsic_est_en;
  a.commit_prev = c->last_M > 0 ? 1 : 0;
  a.scan = c->d_scan, a.perm = c->d_perm;
  fill_quat_const(c, state_point, a.qc);
  a.nfound = c->d_nfound, a.sel = c->d_sel, a.nbr = c->d_nbr, a.far_idx = d_far, a.map_in = c->d_map_in;
  a.ny = c->d_ny, a.trace = c->d_trace, a.wny = d_wny;
  a.cov_threshold = c->prm.cov_threshold, a.fs = c->prm.filter_size_map;
  a.addf = d_addf, a.nonf = d_nonf, a.wp = d_wp;
  hipLaunchKernelGGL(k_mapinc_classify, dim3((N + BLK - 1) / BLK), dim3(BLK), 0, c->stream, a);
  MALIO_HIP(hipGetLastError());  // (no wait: the caller's scans are queued behind it, d_far goes back to the arena in stream order)
  return MALIO_OK;
}

// ---- host side of a pass -------------------------------------------------------------------------
static void q_to_R(const double q[4], double R[9]) { mf::quat_R_eigen(q, R); }

int sums_len(const Ctx *c) { return c->prm.lid_num * NSUM; }

static int nblocks_total(const Ctx *c, int seg_block0[MALIO_MAX_LIDAR + 1]) {
  int b = 0;
  for (int l = 0; l < c->prm.lid_num; l++) {
    seg_block0[l] = b;
    b += (c->seg_start[l + 1] - c->seg_start[l] + BLK - 1) / BLK;
  }
  for (int l = c->prm.lid_num; l <= MALIO_MAX_LIDAR; l++) seg_block0[l] = b;
  return b;
}

int measure_alloc(Ctx *c) {
  size_t N = (size_t)c->N;
  if (N > c->cap_scan) {
    auto fr = [](void *p) {
      if (p) (void)hipFree(p);
    };
    fr(c->d_scan), fr(c->d_perm), fr(c->d_nbr), fr(c->d_plane), fr(c->d_pd2), fr(c->d_world), fr(c->d_ucov),
        fr(c->d_trace), fr(c->d_sel), fr(c->d_nfound), fr(c->d_upload), fr(c->d_world4), fr(c->d_ny), fr(c->d_cert), fr(c->d_kept), fr(c->d_pcache);
    c->cap_scan = N + N / 8 + 1024;
    size_t K = c->cap_scan;
    MALIO_HIP(hipMalloc(&c->d_upload, sizeof(UploadRec) * K));
    MALIO_HIP(hipMalloc(&c->d_scan, sizeof(float4) * K));
    MALIO_HIP(hipMalloc(&c->d_perm, sizeof(u32) * K));
    MALIO_HIP(hipMalloc(&c->d_nbr, sizeof(u32) * 5 * K));
    MALIO_HIP(hipMalloc(&c->d_plane, sizeof(float4) * K));
    MALIO_HIP(hipMalloc(&c->d_pd2, sizeof(float) * K));
    MALIO_HIP(hipMalloc(&c->d_world, sizeof(float) * 3 * K));
    MALIO_HIP(hipMalloc(&c->d_ucov, sizeof(double) * K));
    MALIO_HIP(hipMalloc(&c->d_trace, sizeof(double) * K));
    MALIO_HIP(hipMalloc(&c->d_sel, K));
    MALIO_HIP(hipMalloc(&c->d_nfound, K));
    MALIO_HIP(hipMalloc(&c->d_world4, sizeof(float4) * K));
    MALIO_HIP(hipMalloc(&c->d_ny, sizeof(float) * K));
    MALIO_HIP(hipMalloc(&c->d_cert, sizeof(float4) * K));
    MALIO_HIP(hipMalloc(&c->d_pcache, sizeof(uint4) * K));
    MALIO_HIP(hipMalloc(&c->d_kept, K));
#ifdef MALIO_POISON
    {  // (the per-point state a scan starts from is what its installation writes - scan_install - not what the allocator returned)
      struct { void *p; size_t b; } arr[] = {{c->d_upload, sizeof(UploadRec) * K}, {c->d_scan, sizeof(float4) * K}, {c->d_perm, sizeof(u32) * K},
        {c->d_nbr, sizeof(u32) * 5 * K}, {c->d_plane, sizeof(float4) * K}, {c->d_pd2, sizeof(float) * K}, {c->d_world, sizeof(float) * 3 * K},
        {c->d_ucov, sizeof(double) * K}, {c->d_trace, sizeof(double) * K}, {c->d_sel, K}, {c->d_nfound, K}, {c->d_world4, sizeof(float4) * K},
        {c->d_ny, sizeof(float) * K}, {c->d_cert, sizeof(float4) * K}, {c->d_pcache, sizeof(uint4) * K}, {c->d_kept, K}};
      for (auto &x : arr) MALIO_HIP(hipMemset(x.p, 0xFF, x.b));
      MALIO_HIP(hipDeviceSynchronize());  // (hipMemset may still be queued on the NULL stream, which this handle's stream does not wait for)
    }
#endif
  }
  size_t nb = (N + BLK - 1) / BLK + MALIO_MAX_LIDAR;
  if (nb > c->cap_partials) {
    if (c->d_partials) (void)hipFree(c->d_partials);
    c->cap_partials = nb + nb / 8 + 16;
    MALIO_HIP(hipMalloc(&c->d_partials, sizeof(double) * NSUM * c->cap_partials));  // [NSUM][cap_partials]
  }
  {  // speculating pass (k_pass): one tile per 64 points (+ one per LiDAR for the segment padding), entry-major
    const size_t nt = (N + SQ - 1) / SQ + MALIO_MAX_LIDAR;
    if (nt > c->cap_tiles) {
      if (c->d_tiles) (void)hipFree(c->d_tiles);
      c->cap_tiles = nt + nt / 8 + 16;
      MALIO_HIP(hipMalloc(&c->d_tiles, sizeof(double) * NSUM * c->cap_tiles));
    }
  }
  if (!c->d_mmslots) {
    MALIO_HIP(hipMalloc(&c->d_mmslots, sizeof(u64) * 2 * MM_SLOTS * 5));
    hipLaunchKernelGGL(k_mm_init, dim3(1), dim3(2 * MM_SLOTS), 0, c->stream, c->d_mmslots);
    c->mm_parity = 0;
  }
  if (!c->d_sums) {
    MALIO_HIP(hipMalloc(&c->d_sums, sizeof(double) * (MALIO_MAX_LIDAR * NSUM + 8 + 16)));
    MALIO_HIP(hipHostMalloc(&c->h_sums, sizeof(double) * (MALIO_MAX_LIDAR * NSUM + 8), hipHostMallocDefault));
    MALIO_HIP(hipHostMalloc(&c->h_minmax, sizeof(double) * 8, hipHostMallocDefault));
    // result mailbox of the fused single-GPU pass: the kernels write it, the host reads it after the stream sync
    MALIO_HIP(hipHostMalloc(&c->h_res, sizeof(double) * (MALIO_MAX_LIDAR * NSUM + 16), hipHostMallocMapped | hipHostMallocCoherent));
    MALIO_HIP(hipHostGetDevicePointer((void **)&c->d_res, c->h_res, 0));
  }
  return MALIO_OK;
}

// After a chain of enqueued passes ended out of step with the host's bookkeeping (a gate that gave up): both extrema
// slot sets cleared, parity back to zero, no pending normal_y fold.
int reset_pass_state(Ctx *c) {
  MALIO_HIP(hipStreamSynchronize(c->stream));
  if (c->d_mmslots) hipLaunchKernelGGL(k_mm_init, dim3(1), dim3(2 * MM_SLOTS), 0, c->stream, c->d_mmslots);
  if (c->d_gate_ticket) MALIO_HIP(hipMemsetAsync(c->d_gate_ticket, 0, 256, c->stream));
  MALIO_HIP(hipStreamSynchronize(c->stream));
  c->mm_parity = 0, c->last_M = -1;
  c->mm_guess_valid = false;
  c->cert_valid = false;  // (the chain may have ended between a search pass' kernels: the next search walks every list)
  c->probe_valid = false;
  return MALIO_OK;
}

// ---- once-per-scan grouping of the scan -------------------------------------------------------------------------------
// What the search pass wants from the order of the scan: the points of one LiDAR slot contiguous, slots ascending
// (k_rows_reduce works on per-slot blocks), queries of the same level-1 map cell adjacent (their list reads coalesce and
// hit L1) - and an order that is the same in every run, because every fixed-order reduction over the scan inherits it.
// A full sort delivers that and more (it was one stable radix sort of 32-bit keys: 10 launches of rocprim's merge sort,
// ~0.1 ms for 100 k points, as much as two search passes); nothing needs the cells themselves ordered. So: a bucket
// grouping. key = (slot, cell of the world point under the first pass' state, coordinates modulo 1024: a scan wider
// than 1152 m merely interleaves two far-apart cells); bucket = (slot, column of cells modulo a 64 x 32-column tile, lowest bit of the vertical cell coordinate);
//   k_sort_count    key, bucket, arrival rank inside the bucket (atomic: arbitrary)
//   k_sort_scan     exclusive scan of the bucket counts (one workgroup; clears the counts for the next scan)
//   k_sort_scatter  points to their bucket's segment in arrival order
//   k_sort_place    every point ranks itself inside its segment under (key, index in the caller's cloud) - segments
//                   hold a few points - and goes to segment start + rank: deterministic, same-cell points adjacent;
//                   the kernel writes the sorted scan and resets the per-point state a new scan starts from.
constexpr int SORT_NBK = 4096;                           // buckets per LiDAR slot (100 k points: ~8 per bucket)
constexpr int SORT_NB = SORT_NBK * MALIO_MAX_LIDAR;      // 16384
// counts of k_pack_raw (finished: same stream) -> pinned memory, then the scan's sequence number; called by ONE workgroup
__device__ __forceinline__ void publish_pack(const u32 *__restrict__ info, u32 *pub, u32 seq) {
  if (threadIdx.x < 10) pub[threadIdx.x] = info[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&pub[15], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(64) k_publish_pack(const u32 *__restrict__ info, u32 *pub, u32 seq) { publish_pack(info, pub, seq); }
void publish_pack_now(Ctx *c) {
  if (!c->pack_publish_pending) return;
  c->pack_publish_pending = false;
  hipLaunchKernelGGL(k_publish_pack, dim3(1), dim3(64), 0, c->stream, c->d_packinfo, c->d_packinfo_pub, c->pack_seq);
}
// count_L != 0 (records that arrived packed, malio_scan_set_packed): the points per LiDAR slot and the slots outside
// [0, count_L) are counted HERE instead of by a kernel of their own (a bad slot reads 0 from now on, so that the grouping
// stays inside its buckets; the first pass reports it) and k_sort_scan publishes them.
__global__ void __launch_bounds__(BLK) k_sort_count(UploadRec *in, int n, QuatConst qc, float inv_cf,
                                                    u32 *keys, u32 *bkt, u32 *rnk, u32 *cnt, u32 *pack_info, u32 *pack_pub,
                                                    u32 pack_seq, int count_L, PartView part) {
  if (pack_pub && blockIdx.x == 0) publish_pack(pack_info, pack_pub, pack_seq);  // (uniform per workgroup; under the others' work)
  int i = blockIdx.x * BLK + threadIdx.x;
  if (count_L) {  // (workgroup-uniform)
    __shared__ u32 s_cnt[MALIO_MAX_LIDAR + 1];
    if (threadIdx.x <= MALIO_MAX_LIDAR) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    int lid = -1;
    if (i < n) {
      const u32 w = in[i].w;
      lid = (int)(w & 0xFFu);
      if (lid >= count_L) lid = MALIO_MAX_LIDAR, in[i].w = w & 0xFFFFFF00u;
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int l = 0; l <= MALIO_MAX_LIDAR; l++) {
      const unsigned long long m = __ballot(lid == l);
      if (m && lane == 0) atomicAdd(&s_cnt[l], (u32)__popcll(m));
    }
    __syncthreads();
    if (threadIdx.x <= MALIO_MAX_LIDAR && s_cnt[threadIdx.x])
      atomicAdd(&pack_info[threadIdx.x == MALIO_MAX_LIDAR ? 8 : threadIdx.x], s_cnt[threadIdx.x]);
  }
  if (i >= n) return;
  const UploadRec q = in[i];
  const int lid = (int)(q.w & 0xFF);
  D3 p{(double)q.x, (double)q.y, (double)q.z};
  D3 X = (lid == 0) ? qrot(qc.q0, p) + qc.t0 : qrot(qc.qtc[lid], qrot(qc.ql[lid], p) + qc.tl[lid]) + qc.ttc[lid];
  D3 pg = qrot(qc.rot, X) + qc.pos;
  const u32 cx = (u32)(int)floorf((float)pg.x * inv_cf) & 1023u, cy = (u32)(int)floorf((float)pg.y * inv_cf) & 1023u,
            cz = (u32)(int)floorf((float)pg.z * inv_cf) & 1023u;
  const u32 cell = (cz << 20) | (cy << 10) | cx;
  // bucket = the vertical column of cells, row-major inside a 64 x 64-column tile (72 m at the default edge; farther
  // columns alias, which only interleaves them): neighbouring buckets are neighbouring columns, so consecutive
  // workgroups still work on one part of the map and the five neighbours of their queries share cache lines of the map
  // array. (A hashed bucket order balances the buckets better - the grouping is ~20 us cheaper per scan - but costs
  // every search pass 0.8 us at config 2 and 13 us in config 5's 500 m tunnel; coarser columns change nothing.)
#ifndef KS_BKT_Z
#define KS_BKT_Z 1
#endif
  // KS_BKT_Z bits of the vertical cell coordinate go into the bucket (taken from the horizontal range: 64 x 32 columns): a
  // wall's column of cells is cut into 2^KS_BKT_Z buckets. What a full bucket costs is the same-address atomics below and
  // k_sort_place's quadratic ranking; swept 0 / 1 / 2 / 3 bits (profiles/round4/r04i_grouping_buckets.txt): the update of a
  // new scan 207 / 199.5 / 200 / 201 us at config 2, 434 / 421 / 426 / 421 at config 5, 176 / 172 / 173 / 172 at config 3;
  // the steady search pass does not move.
  constexpr u32 ZB = KS_BKT_Z, XB = 6 - ZB / 2, YB = 6 - (ZB + 1) / 2;
  u32 b = (u32)lid * SORT_NBK + ((((cy & ((1u << YB) - 1u)) << XB) | (cx & ((1u << XB) - 1u))) << ZB) + (cz & ((1u << ZB) - 1u));
  if (part.world > 1) {
    // A tile shard serves the points of its own tiles: the points it owns under THIS state first, then everybody else's,
    // each grouped by tile class (32 classes of the ownership hash) and by 8 x 8-column patch inside the tile, so that the
    // points a shard owns fill whole workgroups instead of a few lanes of most workgroups (the column order above lets a
    // 64-point block run through eight tiles: two thirds of an eighth-shard's workgroups had work) - and fill the FIRST
    // workgroups of every LiDAR segment: a 200 k-point scan is 3 125 workgroups, more than the GPU holds at once, and an
    // owned workgroup of the second generation started microseconds late behind workgroups that had nothing to do.
    // (Ownership is still decided per pass, per point, from that pass' world point - search_wg phase A: a point that
    // changes tiles with the iterate is served by its new owner wherever it sits in this order.)
    const int tx = tile_coord((float)pg.x, part.inv_tile), ty = tile_coord((float)pg.y, part.inv_tile),
              tz = part_tz(part, (float)pg.z);
    const u32 th = tile_hash(tx, ty, tz);
    const u32 other = part_tile_owner(part, tx, ty, tz) == (u32)part.rank ? 0u : 1u;
    b = (u32)lid * SORT_NBK + (other << 11) + (((th >> 8) & 31u) << 6) + (((cy & 7u) << 3) | (cx & 7u));
  }
  keys[i] = cell;
  bkt[i] = b;
  rnk[i] = atomicAdd(&cnt[b], 1u);
}
// one workgroup: 1024 threads x 16 consecutive buckets; thread sums -> wave scan (shuffles) -> scan of the 16 wave totals
__global__ void __launch_bounds__(1024) k_sort_scan(u32 *cnt, u32 *offs, u32 *pack_info, u32 *pack_pub, u32 pack_seq) {  // offs[SORT_NB + 1]
  __shared__ u32 s_wave[16];
  if (pack_pub) {  // counts of k_sort_count<count_L>: to the host, then cleared for the next scan
    publish_pack(pack_info, pack_pub, pack_seq);
    __syncthreads();
    if (threadIdx.x < 16) pack_info[threadIdx.x] = 0u;
  }
  constexpr int PER = SORT_NB / 1024;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  uint4 *c4 = reinterpret_cast<uint4 *>(cnt + (size_t)t * PER);
  uint4 v[PER / 4];
  u32 sum = 0;
#pragma unroll
  for (int k = 0; k < PER / 4; k++) {
    v[k] = c4[k];
    sum += v[k].x + v[k].y + v[k].z + v[k].w;
    c4[k] = make_uint4(0u, 0u, 0u, 0u);
  }
  u32 inc = sum;  // inclusive scan over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_wave[wv] = inc;
  __syncthreads();
  u32 base = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) base += w < wv ? s_wave[w] : 0u;
  u32 run = base + inc - sum;
  u32 *o = offs + (size_t)t * PER;
#pragma unroll
  for (int k = 0; k < PER / 4; k++) {
    o[4 * k] = run, run += v[k].x;
    o[4 * k + 1] = run, run += v[k].y;
    o[4 * k + 2] = run, run += v[k].z;
    o[4 * k + 3] = run, run += v[k].w;
  }
  if (t == 1023) offs[SORT_NB] = run;
}
__global__ void __launch_bounds__(BLK) k_sort_scatter(int n, const u32 *__restrict__ keys, const u32 *__restrict__ bkt,
                                                      const u32 *__restrict__ rnk, const u32 *__restrict__ offs, u64 *tkv,
                                                      u32 *tbkt) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  const u32 b = bkt[i], pos = offs[b] + rnk[i];
  tkv[pos] = ((u64)keys[i] << 32) | (u64)(u32)i, tbkt[pos] = b;  // (cell, index in the caller's cloud): one compare orders them
}
// the sorted scan + the per-scan state every new scan starts from (nothing reads these arrays before the first pass)
__device__ __forceinline__ void scan_install(const UploadRec *__restrict__ in, u32 src, int dst, int n, float4 *out_scan,
                                             u32 *out_perm, float *out_ny, unsigned char *sel, unsigned char *nfound, u32 *nbr,
                                             float *pd2, float4 *plane, float4 *cert) {
  const UploadRec r = in[src];  // src: index in the caller's cloud
  cert[dst] = make_float4(0.f, 0.f, 0.f, 0.f);  // no certificate: the first search pass walks the lists for every point
  out_scan[dst] = make_float4(r.x, r.y, r.z, __uint_as_float(r.w));
  out_perm[dst] = src;
  out_ny[dst] = r.ny;
  sel[dst] = 0, nfound[dst] = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) nbr[(size_t)k * n + dst] = 0xFFFFFFFFu;
  pd2[dst] = 0.f;
  plane[dst] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void __launch_bounds__(BLK) k_sort_place(const UploadRec *__restrict__ in, int n, const u64 *__restrict__ tkv,
                                                    const u32 *__restrict__ tbkt,
                                                    const u32 *__restrict__ offs, float4 *out_scan, u32 *out_perm,
                                                    float *out_ny, unsigned char *sel, unsigned char *nfound, u32 *nbr,
                                                    float *pd2, float4 *plane, float4 *cert) {
  int j = blockIdx.x * BLK + threadIdx.x;
  if (j >= n) return;
  const u32 b = tbkt[j], s0 = offs[b], s1 = offs[b + 1];
  const u64 mine = tkv[j];
  u32 rank = 0;
  // (four loads in flight per step: with a trip count that differs from lane to lane the compiler waits for every load
  // before it issues the next, and the kernel is as long as its fullest bucket's chain of round trips)
  for (u32 m = s0; m < s1; m += 4) {
    u64 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = tkv[min(m + (u32)u, s1 - 1)];
#pragma unroll
    for (int u = 0; u < 4; u++) rank += (m + (u32)u < s1 && v[u] < mine) ? 1u : 0u;
  }
  scan_install(in, (u32)mine, (int)(s0 + rank), n, out_scan, out_perm, out_ny, sel, nfound, nbr, pd2, plane, cert);
}
// the upload order kept (malio_scan_order): install only
__global__ void __launch_bounds__(BLK) k_gather_scan(const UploadRec *__restrict__ in, int n, float4 *out_scan, u32 *out_perm,
                                                     float *out_ny, unsigned char *sel, unsigned char *nfound, u32 *nbr,
                                                     float *pd2, float4 *plane, float4 *cert) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  scan_install(in, (u32)i, i, n, out_scan, out_perm, out_ny, sel, nfound, nbr, pd2, plane, cert);
}

void fill_quat_const(const Ctx *c, const malio_state_t *s, QuatConst &qc) {
  auto Q = [](const double q[4]) { return Q4{q[0], q[1], q[2], q[3]}; };
  auto V = [](const double t[3]) { return D3{t[0], t[1], t[2]}; };
  qc.rot = Q(s->rot), qc.pos = V(s->pos);
  qc.q0 = Q(s->offset_R[0]), qc.t0 = V(s->offset_T[0]);
  for (int l = 0; l < MALIO_MAX_LIDAR; l++) {
    int ll = l < c->prm.lid_num ? l : 0;
    qc.ql[l] = Q(s->offset_R[ll]), qc.tl[l] = V(s->offset_T[ll]);
    if (l >= 1 && l < c->prm.lid_num) {
      qc.qtc[l] = Q(c->tcq[l - 1]), qc.ttc[l] = V(c->tct[l - 1]);
    } else {
      qc.qtc[l] = Q4{0, 0, 0, 1}, qc.ttc[l] = D3{0, 0, 0};
    }
  }
}

// A scan packed on the device (malio_scan_set with a page-locked cloud): its per-slot counts arrive here, once, with the
// first pass - AFTER that pass has queued the grouping and its search kernel, which need none of them: the pack kernel's
// last block left the counts and the scan's sequence number in pinned memory, so this neither copies nor waits for the
// stream (a blocking read-back here used to keep the GPU idle for ~40 us between the upload and the grouping).
int resolve_scan_segments(Ctx *c) {
  if (!c->seg_pending) return MALIO_OK;
  publish_pack_now(c);  // (nobody has yet: the caller wants the counts before any grouping kernel is queued)
  volatile u32 *pub = c->h_packinfo;
  unsigned long long spins = 0;
  while (__atomic_load_n(const_cast<u32 *>(&pub[15]), __ATOMIC_ACQUIRE) != c->pack_seq) {
    if ((++spins & 0x3FFF) == 0) {
      hipError_t q = hipStreamQuery(c->stream);
      if (q != hipErrorNotReady && __atomic_load_n(const_cast<u32 *>(&pub[15]), __ATOMIC_ACQUIRE) != c->pack_seq) {
        MALIO_HIP(q);  // the stream failed; if it merely drained without the word, something is badly wrong
        c->err = "malio_scan_set: the pack kernel ended without publishing its counts";
        return MALIO_ERR_HIP;
      }
    }
    __builtin_ia32_pause();
  }
  u32 info[16];
  for (int k = 0; k < 10; k++) info[k] = pub[k];
  c->seg_pending = false;
  const int L = c->prm.lid_num;
  if (info[8]) {  // int(intensity) outside [0, lid_num): what malio_scan_set rejects on the spot for a pageable cloud
    c->N = 0;
    c->err = "malio_scan_set: a point's LiDAR slot int(intensity) is outside [0, lid_num)";
    return MALIO_ERR_BAD_ARG;
  }
  c->seg_start[0] = 0;
  for (int l = 0; l < MALIO_MAX_LIDAR; l++) c->seg_start[l + 1] = c->seg_start[l] + (l < L ? (int)info[l] : 0);
  if (c->seg_start[L] != c->N) {  // counts that do not add up to the scan would send every per-segment kernel out of bounds
    c->err = "malio_scan_set: the per-slot counts of the scan do not add up to its size (" + std::to_string(c->seg_start[L]) +
             " vs " + std::to_string(c->N) + ")";
    c->N = 0;
    return MALIO_ERR_HIP;
  }
  c->scan_keep_order = c->scan_order_mode == MALIO_SCAN_ORDER_KEEP && info[9] == 0;
  return MALIO_OK;
}

// Grouping of the scan, once per scan, with the first pass' state (coherence only, not results).
static int sort_scan(Ctx *c, const QuatConst &qc) {
  // The temporaries live in the arena and are handed back when this returns, with the kernels still queued: every
  // arena user enqueues on c->stream, so the stream's order is the only synchronisation needed.
  ArenaScope sc(c->arena);
  const int N = c->N;
  const dim3 grid((N + BLK - 1) / BLK);
  if (c->scan_keep_order) {  // malio_scan_order: the upload order is kept (it is grouped by LiDAR slot)
    publish_pack_now(c);
    hipLaunchKernelGGL(k_gather_scan, grid, dim3(BLK), 0, c->stream, c->d_upload, N, c->d_scan, c->d_perm, c->d_ny,
                       c->d_sel, c->d_nfound, c->d_nbr, c->d_pd2, c->d_plane, c->d_cert);
    c->scan_sorted = true;
    return MALIO_OK;
  }
  if (!c->d_sort_cnt) {  // bucket counts (left at zero by every scan) + offsets
    MALIO_HIP(hipMalloc(&c->d_sort_cnt, sizeof(u32) * (2 * SORT_NB + 16)));
    MALIO_HIP(hipMemsetAsync(c->d_sort_cnt, 0, sizeof(u32) * (2 * SORT_NB + 16), c->stream));
  }
  u32 *cnt = c->d_sort_cnt, *offs = c->d_sort_cnt + SORT_NB;
  u32 *keys = nullptr, *bkt = nullptr, *rnk = nullptr, *tbkt = nullptr;
  u64 *tkv = nullptr;
  MALIO_HIP(sc.get(&keys, (size_t)N));
  MALIO_HIP(sc.get(&bkt, (size_t)N));
  MALIO_HIP(sc.get(&rnk, (size_t)N));
  MALIO_HIP(sc.get(&tkv, (size_t)N));
  MALIO_HIP(sc.get(&tbkt, (size_t)N));
  const bool pub = c->pack_publish_pending;
  c->pack_publish_pending = false;
  const bool cis = pub && c->count_in_sort;  // the counts do not exist yet: k_sort_count forms them, k_sort_scan publishes
  c->count_in_sort = false;
  hipLaunchKernelGGL(k_sort_count, grid, dim3(BLK), 0, c->stream, c->d_upload, N, qc, c->inv_cell /* == nl1.inv_cf, which exists only once the lists are built */, keys, bkt, rnk, cnt,
                     c->d_packinfo, pub && !cis ? c->d_packinfo_pub : (u32 *)nullptr, c->pack_seq, cis ? c->prm.lid_num : 0, c->part);
  hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(1024), 0, c->stream, cnt, offs, c->d_packinfo,
                     cis ? c->d_packinfo_pub : (u32 *)nullptr, c->pack_seq);
  hipLaunchKernelGGL(k_sort_scatter, grid, dim3(BLK), 0, c->stream, N, keys, bkt, rnk, offs, tkv, tbkt);
  hipLaunchKernelGGL(k_sort_place, grid, dim3(BLK), 0, c->stream, c->d_upload, N, tkv, tbkt, offs, c->d_scan, c->d_perm,
                     c->d_ny, c->d_sel, c->d_nfound, c->d_nbr, c->d_pd2, c->d_plane, c->d_cert);
  MALIO_HIP(hipGetLastError());
  c->scan_sorted = true;
  return MALIO_OK;
}

// everything of Pass1Args that does not depend on the pass (state, parities and commit_prev are filled by the caller,
// or read from the device loop's control block)
static void fill_pass1_static(Ctx *c, Pass1Args &a) {
  a.N = c->N;
  a.scan = c->d_scan;
  a.map_in = c->d_map_in;
  a.unc = c->d_unc;
  for (int l = 0; l < MALIO_MAX_LIDAR; l++) a.unc_off[l] = c->unc_off[l], a.unc_len[l] = c->unc_len[l];
  a.plane_th = c->prm.plane_th, a.cov_threshold = c->prm.cov_threshold, a.extrinsic_est_en = c->prm.extrinsic_est_en;
  a.world4 = c->d_world4;
  a.part = c->part;
  a.nbr = c->d_nbr, a.plane = c->d_plane, a.pd2 = c->d_pd2, a.world = c->d_world, a.ucov = c->d_ucov;
  a.trace = c->d_trace, a.sel = c->d_sel, a.nfound = c->d_nfound;
  a.ny = c->d_ny;
  a.cert = c->d_cert, a.kept = c->d_kept, a.pcache = c->d_pcache, a.skip = 0;
  a.mm_base = c->d_mmslots;
  a.dl = nullptr, a.mm_cur = a.mm_next = nullptr, a.commit_prev = 0;
}

// matrix form of a state for stage 2 (a5: rotation matrices of the pose, the extrinsics and the temporal compensation)
void fill_pass_const(const Ctx *c, const malio_state_t *s, PassConst &pc) {
  q_to_R(s->rot, pc.Rw);
  q_to_R(s->offset_R[0], pc.R0);
  for (int k = 0; k < 3; k++) pc.pw[k] = s->pos[k], pc.t0[k] = s->offset_T[0][k];
  for (int l = 0; l < MALIO_MAX_LIDAR; l++) {
    int ll = l < c->prm.lid_num ? l : 0;
    q_to_R(s->offset_R[ll], pc.lid[l].Rl);
    for (int k = 0; k < 3; k++) pc.lid[l].tl[k] = s->offset_T[ll][k];
    if (l >= 1 && l < c->prm.lid_num) {
      q_to_R(c->tcq[l - 1], pc.lid[l].Rtc);
      for (int k = 0; k < 3; k++) pc.lid[l].ttc[k] = c->tct[l - 1][k];
    } else {
      const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      memcpy(pc.lid[l].Rtc, I, sizeof(I));
      pc.lid[l].ttc[0] = pc.lid[l].ttc[1] = pc.lid[l].ttc[2] = 0;
    }
  }
  pc.L = c->prm.lid_num, pc.extrinsic_est_en = c->prm.extrinsic_est_en;
  pc.plane_th = c->prm.plane_th, pc.cov_threshold = c->prm.cov_threshold;
}

// Bookkeeping of a SEARCH pass about to be queued: may it keep cached neighbours (search_wg, phase A')? Yes when an
// earlier search pass of THIS scan has left its certificates and the map has not changed since (map ids in d_nbr, and
// the "no outsider inside r" statement, belong to one epoch of the map array). From here on the certificates of this
// pass exist in stream order.
// Bit 1 of the result: the probe cache (Pass1Args::pcache) holds entries of an earlier search pass of this scan against the
// lists as they are (every change of the lists goes with a change of the map array); without it the pass ignores the cache.
// Bit 2: the pass writes the probes it makes (MALIO_OPT_PROBE_CACHE on).
int search_skip_begin(Ctx *c) {
  const int skip = (c->opt_search_skip && c->cert_valid && c->nbr_epoch == c->map_epoch) ? 1 : 0;
  const int probes = c->opt_probe_cache ? ((c->probe_valid && c->nbr_epoch == c->map_epoch) ? 6 : 4) : 0;  // 4: this pass fills the cache
  c->nbr_epoch = c->map_epoch;
  c->cert_valid = c->opt_search_skip != 0;  // (only the SKIP kernels leave certificates)
  c->probe_valid = c->opt_probe_cache != 0;
  c->last_search_skip = skip;
  return skip | probes;
}

int pass_stage1(Ctx *c, const malio_state_t *s, int converge, double *d_minmax4_out) {
  if (c->map_n - c->map_dead <= 0) return MALIO_ERR_NO_MAP;
  if (c->N <= 0) return MALIO_ERR_NO_SCAN;
  Pass1Args a;
  fill_quat_const(c, s, a.qc);
  if (!c->scan_sorted) {
    // (the grouping needs the counts only to decide whether the caller's order can be kept). BEFORE map_sync_search: it
    // reads neither the map nor the lists, so it is queued - and runs - while the previous scan's list maintenance is
    // still busy on its own stream; the search kernels join behind that (maint_join in map_sync_search).
    if (c->scan_order_mode == MALIO_SCAN_ORDER_KEEP && !c->count_in_sort)  // (a packed scan set under SORT is sorted: its counts come from the grouping)
      if (int rc = resolve_scan_segments(c)) return rc;
    int rc = sort_scan(c, a.qc);
    if (rc != MALIO_OK) return rc;
  }
  if (int rc = map_sync_search(c)) return rc;
  fill_pass1_static(c, a);
  c->mm_parity ^= 1;  // this pass accumulates into one parity and clears the other for the next pass
  a.mm_cur = c->d_mmslots + (size_t)c->mm_parity * MM_SLOTS * 5;
  a.mm_next = c->d_mmslots + (size_t)(c->mm_parity ^ 1) * MM_SLOTS * 5;
  a.commit_prev = c->last_M > 0 ? 1 : 0;
  c->last_M = -1;  // the fold is done by this pass; finish_host sets the new value
  const int nb = (c->N + BLK - 1) / BLK;
  c->last_pass_search = converge != 0;
  if (converge) {
    a.skip = search_skip_begin(c);
    const auto kern = c->opt_search_skip ? &k_search<false, true> : &k_search<false, false>;
    hipLaunchKernelGGL(kern, dim3((c->N + SQ - 1) / SQ), dim3(KS_BLK), 0, c->stream, a, view_l1(c), view_of(c->nl2));
    prof_mark(c, "k_search");
  } else {
    hipLaunchKernelGGL(k_reuse, dim3(nb), dim3(BLK), 0, c->stream, a);
    prof_mark(c, "k_reuse");
  }
  if (d_minmax4_out) {  // staged (multi-GPU) path: the caller all-reduces these between the stages
    hipLaunchKernelGGL(k_minmax_reduce, dim3(1), dim3(64), 0, c->stream, (const u64 *)a.mm_cur, c->prm.extrinsic_est_en, d_minmax4_out);
    prof_mark(c, "k_minmax_reduce");
  }
  MALIO_HIP(hipGetLastError());
  fill_pass_const(c, s, c->pc);  // matrix form of the same state for stage 2
  return resolve_scan_segments(c);  // stage 2 is launched per LiDAR segment
}

static int fill_pass2_static(Ctx *c, Pass2Args &a) {
  a.N = c->N, a.L = c->prm.lid_num, a.extrinsic_est_en = c->prm.extrinsic_est_en;
  a.scan = c->d_scan, a.plane = c->d_plane, a.pd2 = c->d_pd2, a.ucov = c->d_ucov, a.trace = c->d_trace, a.sel = c->d_sel;
  int nb = nblocks_total(c, a.seg_block0);
  for (int l = 0; l <= MALIO_MAX_LIDAR; l++) a.seg_start[l] = c->seg_start[l < c->prm.lid_num ? l : c->prm.lid_num];
  a.wc.plane_cov_max = c->prm.plane_cov_max, a.wc.plane_cov_min = c->prm.plane_cov_min;
  a.wc.point_cov_max = c->prm.point_cov_max, a.wc.point_cov_min = c->prm.point_cov_min;
  a.wc.range_min = c->prm.range_min, a.wc.range_max = c->prm.range_max;
  a.partials = c->d_partials, a.pstride = (int)c->cap_partials;
  a.rows = nullptr, a.minmax4 = nullptr, a.mmslots = nullptr, a.mm_out = nullptr;
  a.dl = nullptr, a.mm_base = c->d_mmslots;
  return nb;
}

int pass_stage2(Ctx *c, const double *d_minmax4_in, double *d_mm_out, double *d_sums_out, bool want_rows, const GateArgs *gate) {
  Pass2Args a;
  const int nb = fill_pass2_static(c, a);
  for (int l = 0; l < c->prm.lid_num; l++)  // k_final_reduce<4>: 64 rounds of 256 workgroup partials per LiDAR segment
    if ((c->seg_start[l + 1] - c->seg_start[l] + BLK - 1) / BLK > 64 * 256) {
      c->err = "malio_measure: more than 4 194 304 scan points in one LiDAR slot";
      return MALIO_ERR_BAD_ARG;
    }
  a.pc = c->pc;
  a.minmax4 = d_minmax4_in;
  a.mmslots = c->d_mmslots + (size_t)c->mm_parity * MM_SLOTS * 5, a.mm_out = d_mm_out;
  if (want_rows) {
    size_t need = (size_t)c->N * 14;
    if (need > c->cap_rows) {
      if (c->d_rows) (void)hipFree(c->d_rows);
      c->cap_rows = need + need / 8;
      MALIO_HIP(hipMalloc(&c->d_rows, sizeof(double) * c->cap_rows));
    }
    a.rows = c->d_rows;
  }
  hipLaunchKernelGGL(k_rows_reduce<false>, dim3(nb), dim3(BLK), 0, c->stream, a);
  prof_mark(c, "k_rows_reduce");
  SegBlocks sb;
  for (int l = 0; l <= MALIO_MAX_LIDAR; l++) sb.b[l] = a.seg_block0[l];
  hipLaunchKernelGGL(k_final_reduce<4>, dim3((c->prm.lid_num * NSUM * 64 + BLK - 1) / BLK), dim3(BLK), 0, c->stream,
                     c->d_partials, (int)c->cap_partials, sb, c->prm.lid_num, d_sums_out, (const DevLoop *)nullptr,
                     gate ? *gate : GateArgs{}, FoldArgs{});
  prof_mark(c, "k_final_reduce");
  MALIO_HIP(hipGetLastError());
  return MALIO_OK;
}

// Device loop (csrc/ieskf_dev.hip): sort the scan (once per scan) with the state the loop starts from
int prepare_scan_dev(Ctx *c, const malio_state_t *s) {
  if (c->map_n - c->map_dead <= 0) return MALIO_ERR_NO_MAP;
  if (c->N <= 0) return MALIO_ERR_NO_SCAN;
  if (!c->scan_sorted) {  // (before map_sync_search: see pass_stage1)
    if (c->scan_order_mode == MALIO_SCAN_ORDER_KEEP && !c->count_in_sort)  // (a packed scan set under SORT is sorted: its counts come from the grouping)
      if (int rc = resolve_scan_segments(c)) return rc;
    QuatConst qc;
    fill_quat_const(c, s, qc);
    if (int rc = sort_scan(c, qc)) return rc;
  }
  if (int rc = map_sync_search(c)) return rc;
  return MALIO_OK;  // (the segments are resolved by whoever launches a stage 2: pass_stage1, or the device loop)
}

// One pass of the device loop: the same kernels, every pass-dependent input read from c->d_loop. Nothing here depends
// on what the pass will turn out to be: a search pass, a reuse pass (inside k_search<true>) or nothing (loop over).
int enqueue_pass_dev(Ctx *c, double *d_sums_out, double *d_mm_out, const GateArgs *gate) {
  Pass1Args a;
  fill_pass1_static(c, a);
  a.dl = c->d_loop;
  // (whether a search pass of the enqueued-ahead loop may keep neighbours is in the control block; the instantiation that can
  // is used whenever the option is on)
  const auto kern = c->opt_search_skip ? &k_search<true, true> : &k_search<true, false>;
  hipLaunchKernelGGL(kern, dim3((c->N + SQ - 1) / SQ), dim3(KS_BLK), 0, c->stream, a, view_l1(c), view_of(c->nl2));
  Pass2Args b;
  const int nb = fill_pass2_static(c, b);
  b.dl = c->d_loop, b.mm_out = d_mm_out;
  hipLaunchKernelGGL(k_rows_reduce<true>, dim3(nb), dim3(BLK), 0, c->stream, b);
  SegBlocks sb;
  for (int l = 0; l <= MALIO_MAX_LIDAR; l++) sb.b[l] = b.seg_block0[l];
  hipLaunchKernelGGL(k_final_reduce<4>, dim3((c->prm.lid_num * NSUM * 64 + BLK - 1) / BLK), dim3(BLK), 0, c->stream,
                     c->d_partials, (int)c->cap_partials, sb, c->prm.lid_num, d_sums_out, (const DevLoop *)c->d_loop,
                     gate ? *gate : GateArgs{}, FoldArgs{});
  MALIO_HIP(hipGetLastError());
  return MALIO_OK;
}

// ---- host side of the speculating pass ----------------------------------------------------------------------------------
bool fuse_eligible(Ctx *c, int converge, bool need_guess) {
  if (!c->fuse_enabled || (need_guess && !c->mm_guess_valid) || !c->scan_sorted || c->seg_pending) return false;
  if (!c->d_tiles) return false;
  // k_final_reduce<16> keeps one node per round of 1 024 tiles in 64 LDS slots: a LiDAR segment above 64 x 1 024 tiles (4.2 M
  // points) takes the three-kernel pass
  for (int l = 0; l < c->prm.lid_num; l++)
    if ((c->seg_start[l + 1] - c->seg_start[l] + SQ - 1) / SQ > 64 * 1024) return false;
  // A wrong guess costs a pass its rows a second time (and the enqueued-ahead update a whole repeated unit, ~30 us), a
  // right one saves 1-4 us: after a miss the handle stops speculating for a number of eligible passes that grows with the
  // miss rate (fused_collect). Scenes whose extrema move with every iterate (planes of very uneven covariance: BASELINE
  // config 3 misses 8 % of its guesses) then speculate rarely; the usual scene misses one guess in 25 or fewer.
  if (c->fuse_cooldown > 0 && !c->fuse_debug_bad_guess) {
    c->fuse_cooldown--;
    return false;
  }
  return true;
}

// everything of FuseArgs that does not depend on the pass; returns the number of workgroups (= tiles); sb: the segments
// in tiles, for k_final_reduce<16>
static int fill_fuse_static(Ctx *c, FuseArgs &f, SegBlocks &sb) {
  const int L = c->prm.lid_num;
  int b = 0;
  for (int l = 0; l <= MALIO_MAX_LIDAR; l++) {
    const int ll = l < L ? l : L;
    f.seg_start[l] = c->seg_start[ll];
    f.seg_blk0[l] = b, sb.b[l] = b;
    if (l < L) b += (c->seg_start[l + 1] - c->seg_start[l] + SQ - 1) / SQ;
  }
  f.L = L, f.converge = 0;
  f.wc.plane_cov_max = c->prm.plane_cov_max, f.wc.plane_cov_min = c->prm.plane_cov_min;
  f.wc.point_cov_max = c->prm.point_cov_max, f.wc.point_cov_min = c->prm.point_cov_min;
  f.wc.range_min = c->prm.range_min, f.wc.range_max = c->prm.range_max;
  f.tiles = c->d_tiles, f.tstride = (int)c->cap_tiles;
  memset(&f.pc, 0, sizeof(f.pc));
  memset(f.guess, 0, sizeof(f.guess));
  return b;
}

// row (optional): where [sums | extrema words] go - the handle's pinned result buffer, or the device row of an exchange
static void launch_final_tiles(Ctx *c, const SegBlocks &sb, const DevLoop *dl, const GateArgs *gate, double *row = nullptr) {
  const int ns = sums_len(c);
  if (!row) row = c->d_res;
  FoldArgs fold;
  fold.mmslots = dl ? c->d_mmslots : c->d_mmslots + (size_t)c->mm_parity * MM_SLOTS * 5;
  fold.mm_out = row + ns, fold.extrinsic_est_en = c->prm.extrinsic_est_en;
  hipLaunchKernelGGL(k_final_reduce<16>, dim3((c->prm.lid_num * NSUM + FR16_BLK / 256 - 1) / (FR16_BLK / 256)), dim3(FR16_BLK), 0, c->stream,
                     (const double *)c->d_tiles, (int)c->cap_tiles, sb, c->prm.lid_num, row, dl, gate ? *gate : GateArgs{}, fold);
}

// one pass as k_pass -> k_final_reduce<16>, state in the kernel arguments (malio_measure, the host-driven loop): the
// bookkeeping of pass_stage1 + the launches. Results land in h_res like the three-kernel pass'. The caller has checked
// fuse_eligible.
int pass_fused(Ctx *c, const malio_state_t *s, int converge, const GateArgs *gate, double *row) {
  if (c->map_n - c->map_dead <= 0) return MALIO_ERR_NO_MAP;
  if (c->N <= 0) return MALIO_ERR_NO_SCAN;
  if (int rc = map_sync_search(c)) return rc;
  Pass1Args a;
  fill_quat_const(c, s, a.qc);
  fill_pass1_static(c, a);
  c->mm_parity ^= 1;
  a.mm_cur = c->d_mmslots + (size_t)c->mm_parity * MM_SLOTS * 5;
  a.mm_next = c->d_mmslots + (size_t)(c->mm_parity ^ 1) * MM_SLOTS * 5;
  a.commit_prev = c->last_M > 0 ? 1 : 0;
  c->last_M = -1;
  c->last_pass_search = converge != 0;
  if (converge) a.skip = search_skip_begin(c);
  if (!converge && KS_REUSE_ROWS) {  // the streaming form (k_reuse_rows): k_rows_reduce's workgroups, partials and final reduction
    Pass2Args p2;
    const int nb = fill_pass2_static(c, p2);
    bool fits = true;
    for (int l = 0; l < c->prm.lid_num; l++)  // (k_final_reduce<4>: 64 rounds of 256 partials per LiDAR segment; longer ones take k_pass)
      if ((c->seg_start[l + 1] - c->seg_start[l] + BLK - 1) / BLK > 64 * 256) fits = false;
    if (fits) {
      ReuseRowsArgs f;
      for (int l = 0; l <= MALIO_MAX_LIDAR; l++) f.seg_block0[l] = p2.seg_block0[l], f.seg_start[l] = p2.seg_start[l];
      f.L = c->prm.lid_num, f.wc = p2.wc;
      fill_pass_const(c, s, c->pc);
      f.pc = c->pc;
      memcpy(f.guess, c->mm_guess, sizeof(f.guess));
      if (c->fuse_debug_bad_guess) f.guess[0] += 1.0;
      memcpy(c->fuse_guess_used, f.guess, sizeof(f.guess));
      f.partials = c->d_partials, f.pstride = (int)c->cap_partials;
      hipLaunchKernelGGL(k_reuse_rows, dim3(nb), dim3(BLK), 0, c->stream, a, f);
      prof_mark(c, "k_reuse_rows");
      SegBlocks sb;
      for (int l = 0; l <= MALIO_MAX_LIDAR; l++) sb.b[l] = p2.seg_block0[l];
      const int ns = sums_len(c);
      double *out_row = row ? row : c->d_res;
      FoldArgs fold;
      fold.mmslots = c->d_mmslots + (size_t)c->mm_parity * MM_SLOTS * 5, fold.mm_out = out_row + ns, fold.extrinsic_est_en = c->prm.extrinsic_est_en;
      hipLaunchKernelGGL(k_final_reduce<4>, dim3((c->prm.lid_num * NSUM * 64 + BLK - 1) / BLK), dim3(BLK), 0, c->stream,
                         (const double *)c->d_partials, (int)c->cap_partials, sb, c->prm.lid_num, out_row, (const DevLoop *)nullptr,
                         gate ? *gate : GateArgs{}, fold);
      prof_mark(c, "k_final_reduce");
      MALIO_HIP(hipGetLastError());
      c->fuse_passes++;
      return MALIO_OK;
    }
  }
  FuseArgs f;
  SegBlocks sb;
  const int nwg = fill_fuse_static(c, f, sb);
  f.converge = converge;
  fill_pass_const(c, s, c->pc);
  f.pc = c->pc;
  memcpy(f.guess, c->mm_guess, sizeof(f.guess));
  if (c->fuse_debug_bad_guess) f.guess[0] += 1.0;
  memcpy(c->fuse_guess_used, f.guess, sizeof(f.guess));
  const auto kern = c->opt_search_skip ? &k_pass<false, true> : &k_pass<false, false>;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(KS_BLK), 0, c->stream, a, view_l1(c), view_of(c->nl2), f, (const DevLoop *)nullptr);
  prof_mark(c, "k_pass");
  launch_final_tiles(c, sb, nullptr, gate, row);
  prof_mark(c, "k_final_reduce");
  MALIO_HIP(hipGetLastError());
  c->fuse_passes++;
  return MALIO_OK;
}

// the same two kernels reading state, pass kind, parities and the guess from the device loop's control block
int enqueue_pass_fused_dev(Ctx *c, const GateArgs *gate) {
  Pass1Args a;
  fill_pass1_static(c, a);
  memset(&a.qc, 0, sizeof(a.qc));
  FuseArgs f;
  SegBlocks sb;
  const int nwg = fill_fuse_static(c, f, sb);
  const auto kern = c->opt_search_skip ? &k_pass<true, true> : &k_pass<true, false>;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(KS_BLK), 0, c->stream, a, view_l1(c), view_of(c->nl2), f, (const DevLoop *)c->d_loop);
  launch_final_tiles(c, sb, c->d_loop, gate);
  MALIO_HIP(hipGetLastError());
  c->fuse_passes++;
  return MALIO_OK;
}

void fuse_note(Ctx *c, bool hit);
// the pass' results are in h_res (sums | true extrema): did the guess hold?
int fused_collect(Ctx *c, double *sums_out, bool *hit) {
  const int ns = sums_len(c);
  (void)sums_out;
  *hit = memcmp(c->h_res + ns, c->fuse_guess_used, sizeof(double) * 4) == 0;
  fuse_note(c, *hit);
  return MALIO_OK;
}
// bookkeeping of one speculating pass (also called by malio_measure_node, whose hit / miss is decided across the shards)
void fuse_note(Ctx *c, bool hit_) {
  const bool *hit = &hit_;
  // A miss stops the speculation for the rest of the update at first (3 passes); every further miss before 16 hits in a
  // row doubles that, up to FUSE_COOLDOWN_MAX: a scene that misses one guess in 25 keeps speculating (a hit is worth
  // ~2.5 us, a miss ~25 us: break-even at one in 11), one that misses one in 12 (BASELINE config 3) soon stops.
  if (*hit) {
    c->fuse_hits++;
    if (++c->fuse_hits_in_row >= 16) c->fuse_cooldown_len = FUSE_COOLDOWN_MIN;
  } else {
    c->fuse_misses++, c->fuse_hits_in_row = 0;
    c->fuse_cooldown = c->fuse_cooldown_len;
    c->fuse_cooldown_len = std::min(2 * c->fuse_cooldown_len, FUSE_COOLDOWN_MAX);
  }
}

// Assemble the C x C normal equations from the per-LiDAR 12 x 12 blocks, apply the localization
// weight (laserMapping.cpp:745-759). sums: [L][NSUM]; minmax4: [max_u, -min_u, max_R, -min_R].
int finish_host(Ctx *c, const double *sums, const double *minmax4, malio_measure_out_t *out) {
  const int L = c->prm.lid_num, C = 6 * (1 + L);
  double NtN[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double M = 0;
  memset(out->HtRinvH, 0, sizeof(out->HtRinvH));
  memset(out->HtRinvh, 0, sizeof(out->HtRinvh));
  for (int l = 0; l < L; l++) {
    const double *s = sums + (size_t)l * NSUM;
    int gi[12];
    for (int k = 0; k < 6; k++) gi[k] = k;
    for (int k = 0; k < 3; k++) gi[6 + k] = 6 + 3 * l + k, gi[9 + k] = 6 + 3 * (L + l) + k;
    int e = 0;
    for (int a = 0; a < 12; a++)
      for (int b = a; b < 12; b++, e++) {
        out->HtRinvH[gi[a] * C + gi[b]] += s[e];
        if (a != b) out->HtRinvH[gi[b] * C + gi[a]] += s[e];
      }
    for (int a = 0; a < 12; a++) out->HtRinvh[gi[a]] += s[78 + a];
    NtN[0][0] += s[90], NtN[1][1] += s[91], NtN[2][2] += s[92];
    NtN[0][1] += s[93], NtN[0][2] += s[94], NtN[1][2] += s[95];
    M += s[96];
  }
  NtN[1][0] = NtN[0][1], NtN[2][0] = NtN[0][2], NtN[2][1] = NtN[1][2];
  out->M = (int)(M + 0.5);
  out->unit_cov_minmax[0] = -minmax4[1], out->unit_cov_minmax[1] = minmax4[0];
  out->R_minmax[0] = -minmax4[3], out->R_minmax[1] = minmax4[2];
  if (out->M < 1) {  // laserMapping.cpp:635-639
    out->valid = 0;
    out->w_loc = 0;
    return MALIO_NO_EFFECTIVE_POINTS;
  }
  out->valid = 1;
  const double weight = mf::localize_weight(NtN[0][0], NtN[1][1], NtN[2][2], NtN[0][1], NtN[0][2], NtN[1][2],
                                            c->prm.localize_thresh_min, c->prm.localize_thresh_max,
                                            c->prm.localize_cov_min, c->prm.localize_cov_max);
  out->w_loc = weight;
  const double w2 = weight * weight;
  for (int k = 0; k < C * C; k++) out->HtRinvH[k] *= w2;
  for (int k = 0; k < C; k++) out->HtRinvh[k] *= w2;
  return MALIO_OK;
}

}  // namespace malio
