// Exchange of the per-pass results between the ranks of ONE node through POSIX shared memory.
// What the ranks exchange per pass is 2.4 KB that the HOST needs (the n x n filter algebra runs there, on every rank,
// on the same reduced sums): [L x 97 local sums | 8 extrema words]. Through a GPU collective that is a device
// all-gather of a latency-bound message plus a copy back to the host; through shared memory it is one cache-line
// hand-off per rank. The ranks still hold the map and run the kernels on their own GPU; nothing but these rows moves.
// (Ranks on different nodes cannot use this; dist.py then falls back to the collective.)
//
// Layout of the segment: world x 64 B sequence words | 2 x world x row doubles (two buffers alternating by epoch).
// all_gather(e): write own row into buffer e & 1, publish seq[rank] = e (release), wait until every seq[r] >= e
// (acquire), read all rows. Two buffers suffice: a rank can only start epoch e + 2 after every rank published e + 1,
// which each does after it has finished reading epoch e.
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <memory>
#include "../../include/malio.h"

// Three carriers behind one interface (all_gather of one row per rank, rows back in rank order):
//   SHM    ranks = processes of one node, rows in a POSIX shared-memory segment
//   LOCAL  ranks = threads of one process (the node handle, host/node.cpp), rows in a heap block: same protocol
//   RCCL   ranks = anything RCCL connects (one process per GPU over xGMI, or the threads of a node handle): the row is
//          produced in HBM, ncclAllGather moves it on the handle's stream, one copy brings all rows to pinned memory
enum { XCHG_SHM = 0, XCHG_LOCAL = 1, XCHG_RCCL = 2 };

struct malio_xchg {
  int kind = XCHG_SHM;
  int rank = 0, world = 0, row = 0;
  bool owner = false;
  size_t bytes = 0;
  char *base = nullptr;
  std::shared_ptr<std::vector<char>> heap;  // LOCAL: the block all ranks share
  uint64_t epoch = 0;
  std::string name;
  // RCCL
  ncclComm_t comm = nullptr;
  int device = -1;
  double *d_row = nullptr, *d_all = nullptr, *h_all = nullptr;  // [row], [world][row] in HBM, [world][row] pinned
  std::vector<double> all;  // [world][row] scratch of malio_xchg_reduce
  std::atomic<uint64_t> *seq(int r) const { return reinterpret_cast<std::atomic<uint64_t> *>(base + (size_t)r * 64); }
  double *data(int buf, int r) const {
    return reinterpret_cast<double *>(base + (size_t)world * 64) + ((size_t)buf * world + r) * row;
  }
};

extern "C" {

int malio_xchg_create(const char *name, int rank, int world, int row_doubles, int create, malio_xchg_t *out) {
  if (!name || name[0] != '/' || !out || world < 1 || rank < 0 || rank >= world || row_doubles < 1) return MALIO_ERR_BAD_ARG;
  *out = nullptr;
  malio_xchg *x = new (std::nothrow) malio_xchg();
  if (!x) return MALIO_ERR_ALLOC;
  x->rank = rank, x->world = world, x->row = row_doubles, x->owner = create != 0, x->name = name;
  x->bytes = (size_t)world * 64 + sizeof(double) * 2 * (size_t)world * row_doubles;
  int fd = create ? shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600) : shm_open(name, O_RDWR, 0600);
  if (fd < 0 && create && errno == EEXIST) {  // left behind by a run that died: the name belongs to this job now
    shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
  }
  if (fd < 0) {
    delete x;
    return MALIO_ERR_ALLOC;
  }
  if (create && ftruncate(fd, (off_t)x->bytes) != 0) {
    close(fd);
    shm_unlink(name);
    delete x;
    return MALIO_ERR_ALLOC;
  }
  if (!create) {  // the creator sized it before anybody else was told the name
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < x->bytes) {
      close(fd);
      delete x;
      return MALIO_ERR_BAD_ARG;
    }
  }
  void *p = mmap(nullptr, x->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    if (create) shm_unlink(name);
    delete x;
    return MALIO_ERR_ALLOC;
  }
  x->base = (char *)p;
  if (create) std::memset(x->base, 0, x->bytes);
  *out = x;
  return MALIO_OK;
}

int malio_xchg_create_local(int world, int row_doubles, malio_xchg_t *out_world) {
  if (!out_world || world < 1 || row_doubles < 1) return MALIO_ERR_BAD_ARG;
  const size_t bytes = (size_t)world * 64 + sizeof(double) * 2 * (size_t)world * row_doubles;
  auto heap = std::make_shared<std::vector<char>>(bytes + 64, 0);
  char *base = heap->data() + ((64 - (reinterpret_cast<uintptr_t>(heap->data()) & 63)) & 63);  // sequence words on own lines
  for (int r = 0; r < world; r++) {
    malio_xchg *x = new (std::nothrow) malio_xchg();
    if (!x) {
      for (int k = 0; k < r; k++) delete out_world[k];
      return MALIO_ERR_ALLOC;
    }
    x->kind = XCHG_LOCAL, x->rank = r, x->world = world, x->row = row_doubles, x->bytes = bytes, x->base = base, x->heap = heap;
    out_world[r] = x;
  }
  return MALIO_OK;
}

int malio_rccl_unique_id(void *out128) {
  if (!out128) return MALIO_ERR_BAD_ARG;
  static_assert(sizeof(ncclUniqueId) == MALIO_RCCL_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return MALIO_ERR_HIP;
  std::memcpy(out128, &id, sizeof(id));
  return MALIO_OK;
}

static int rccl_buffers(malio_xchg *x) {
  const size_t rb = sizeof(double) * (size_t)x->row;
  if (hipMalloc((void **)&x->d_row, rb) != hipSuccess || hipMalloc((void **)&x->d_all, rb * x->world) != hipSuccess ||
      hipHostMalloc((void **)&x->h_all, rb * x->world, hipHostMallocDefault) != hipSuccess)
    return MALIO_ERR_ALLOC;
  (void)hipMemset(x->d_row, 0, rb);
  return MALIO_OK;
}

int malio_xchg_create_rccl(const void *unique_id128, int rank, int world, int row_doubles, int device, malio_xchg_t *out) {
  if (!unique_id128 || !out || world < 1 || rank < 0 || rank >= world || row_doubles < 1) return MALIO_ERR_BAD_ARG;
  *out = nullptr;
  if (hipSetDevice(device) != hipSuccess) return MALIO_ERR_NO_DEVICE;
  malio_xchg *x = new (std::nothrow) malio_xchg();
  if (!x) return MALIO_ERR_ALLOC;
  x->kind = XCHG_RCCL, x->rank = rank, x->world = world, x->row = row_doubles, x->device = device;
  ncclUniqueId id;
  std::memcpy(&id, unique_id128, sizeof(id));
  if (ncclCommInitRank(&x->comm, world, id, rank) != ncclSuccess) {
    delete x;
    return MALIO_ERR_HIP;
  }
  if (int rc = rccl_buffers(x)) {
    malio_xchg_destroy(x);
    return rc;
  }
  *out = x;
  return MALIO_OK;
}

int malio_xchg_kind(malio_xchg_t x) { return x ? x->kind : -1; }

int malio_xchg_device_row(malio_xchg_t x, double **d_row) {
  if (!x || !d_row || x->kind != XCHG_RCCL) return MALIO_ERR_BAD_ARG;
  *d_row = x->d_row;
  return MALIO_OK;
}

// RCCL: all ranks' rows, gathered from x->d_row on `stream` (the stream the producing kernels were queued on: the
// collective needs no host synchronisation before it), copied to pinned memory, ONE synchronisation at the end
static int rccl_gather(malio_xchg *x, void *stream, double *out_all) {
  hipStream_t st = (hipStream_t)stream;
  if (ncclAllGather(x->d_row, x->d_all, (size_t)x->row, ncclDouble, x->comm, st) != ncclSuccess) return MALIO_ERR_HIP;
  const size_t bytes = sizeof(double) * (size_t)x->row * x->world;
  if (hipMemcpyAsync(x->h_all, x->d_all, bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return MALIO_ERR_HIP;
  if (hipStreamSynchronize(st) != hipSuccess) return MALIO_ERR_HIP;
  std::memcpy(out_all, x->h_all, bytes);
  return MALIO_OK;
}

int malio_xchg_all_gather(malio_xchg_t x, const double *in, double *out_all, double timeout_s) {
  if (!x || !in || !out_all) return MALIO_ERR_BAD_ARG;
  if (x->kind == XCHG_RCCL) {  // host row in, host rows out (tests, bring-up): staged through the device row
    if (hipSetDevice(x->device) != hipSuccess) return MALIO_ERR_HIP;
    if (hipMemcpy(x->d_row, in, sizeof(double) * (size_t)x->row, hipMemcpyHostToDevice) != hipSuccess) return MALIO_ERR_HIP;
    return rccl_gather(x, nullptr, out_all);
  }
  const uint64_t e = ++x->epoch;
  const int buf = (int)(e & 1);
  std::memcpy(x->data(buf, x->rank), in, sizeof(double) * x->row);
  x->seq(x->rank)->store(e, std::memory_order_release);
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < x->world; r++) {
    unsigned spins = 0;
    while (x->seq(r)->load(std::memory_order_acquire) < e) {
      if (++spins < 4096) {
        __builtin_ia32_pause();
        continue;
      }
      spins = 0;
      sched_yield();  // oversubscribed hosts (tests: several ranks per core) must not livelock
      if (timeout_s > 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
        return MALIO_ERR_TIMEOUT;  // a rank died or fell out of step: fail instead of hanging the node
    }
  }
  std::memcpy(out_all, x->data(buf, 0), sizeof(double) * (size_t)x->world * x->row);
  return MALIO_OK;
}

static int reduce_gathered(malio_xchg *x, int ns, const double *guess4, double *sums_out, double *extrema4_out);

int malio_xchg_reduce_stream(malio_xchg_t x, void *stream, int ns, const double *guess4, double *sums_out,
                             double *extrema4_out, double *own_words_out) {
  if (!x || x->kind != XCHG_RCCL || !sums_out || !extrema4_out || ns < 0 || ns + 4 > x->row) return MALIO_ERR_BAD_ARG;
  x->all.resize((size_t)x->world * x->row);
  int rc = rccl_gather(x, stream, x->all.data());
  if (rc != MALIO_OK) return rc;
  if (own_words_out)  // this rank's own words after the four extrema (count, search diagnostic, ...)
    std::memcpy(own_words_out, &x->all[(size_t)x->rank * x->row + ns + 4], sizeof(double) * (size_t)(x->row - ns - 4));
  return reduce_gathered(x, ns, guess4, sums_out, extrema4_out);
}

int malio_xchg_reduce(malio_xchg_t x, const double *row_in, int ns, const double *guess4, double *sums_out,
                      double *extrema4_out, double timeout_s) {
  if (!x || !row_in || !sums_out || !extrema4_out || ns < 0 || ns + 4 > x->row) return MALIO_ERR_BAD_ARG;
  x->all.resize((size_t)x->world * x->row);
  int rc = malio_xchg_all_gather(x, row_in, x->all.data(), timeout_s);
  if (rc != MALIO_OK) return rc;
  return reduce_gathered(x, ns, guess4, sums_out, extrema4_out);
}

static int reduce_gathered(malio_xchg *x, int ns, const double *guess4, double *sums_out, double *extrema4_out) {
  double E[4];
  for (int k = 0; k < 4; k++) {
    E[k] = x->all[(size_t)ns + k];
    for (int r = 1; r < x->world; r++) {
      const double v = x->all[(size_t)r * x->row + ns + k];
      if (v > E[k]) E[k] = v;
    }
  }
  const bool miss = guess4 && std::memcmp(E, guess4, sizeof(E)) != 0;  // bitwise: the rows were weighted with guess4
  std::memcpy(extrema4_out, E, sizeof(E));
  if (miss) return 1;
  for (int e = 0; e < ns; e++) {  // rank order: every rank forms the same bits
    double s = x->all[e];
    for (int r = 1; r < x->world; r++) s += x->all[(size_t)r * x->row + e];
    sums_out[e] = s;
  }
  return MALIO_OK;
}

int malio_xchg_row(malio_xchg_t x) { return x ? x->row : 0; }
}  // extern "C"
namespace malio {
// did every rank send the same value in word `word` of the rows gathered last? (malio_measure_node: the update-loop mode word)
bool xchg_word_agrees(malio_xchg_t x, int word) {
  if (!x || word < 0 || word >= x->row || x->all.size() < (size_t)x->world * x->row) return true;
  for (int r = 1; r < x->world; r++)
    if (std::memcmp(&x->all[(size_t)r * x->row + word], &x->all[word], sizeof(double)) != 0) return false;
  return true;
}
}  // namespace malio
extern "C" {

// Diagnostics: the latency of one malio_xchg_reduce between `world` native threads of this process on host memory (what
// the node handle's MALIO_NODE_XCHG_HOST exchange costs a pass, without any GPU work around it): median-free mean over
// `iters` lock-step rounds after a warm-up, the slowest thread's figure, in microseconds.
int malio_debug_xchg_latency(int world, int row_doubles, int iters, double *us_out) {
  if (world < 1 || world > 64 || row_doubles < 9 || iters < 1 || !us_out) return MALIO_ERR_BAD_ARG;
  std::vector<malio_xchg_t> xs(world, nullptr);
  if (malio_xchg_create_local(world, row_doubles, xs.data()) != MALIO_OK) return MALIO_ERR_ALLOC;
  std::vector<double> us(world, 0.0);
  std::vector<int> rcs(world, 0);
  std::vector<std::thread> th;
  for (int r = 0; r < world; r++)
    th.emplace_back([&, r] {
      std::vector<double> row(row_doubles), E(4);
      for (int k = 0; k < row_doubles; k++) row[k] = k + r;
      const int ns = row_doubles - 8;
      for (int k = 0; k < 200 && rcs[r] >= 0; k++) rcs[r] = malio_xchg_reduce(xs[r], row.data(), ns, nullptr, row.data(), E.data(), 20.0);
      const auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < iters && rcs[r] >= 0; k++) rcs[r] = malio_xchg_reduce(xs[r], row.data(), ns, nullptr, row.data(), E.data(), 20.0);
      us[r] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
    });
  for (auto &t : th) t.join();
  double worst = 0;
  int rc = MALIO_OK;
  for (int r = 0; r < world; r++) {
    worst = std::max(worst, us[r]);
    if (rcs[r] < 0) rc = rcs[r];
    malio_xchg_destroy(xs[r]);
  }
  *us_out = worst;
  return rc;
}

int malio_xchg_unlink(malio_xchg_t x) {  // once every rank has opened the segment its name is no longer needed
  if (!x) return MALIO_ERR_BAD_ARG;
  if (x->owner) {
    shm_unlink(x->name.c_str());
    x->owner = false;
  }
  return MALIO_OK;
}

int malio_xchg_destroy(malio_xchg_t x) {
  if (!x) return MALIO_OK;
  if (x->kind == XCHG_RCCL) {
    if (x->device >= 0) (void)hipSetDevice(x->device);
    if (x->comm) (void)ncclCommDestroy(x->comm);
    if (x->d_row) (void)hipFree(x->d_row);
    if (x->d_all) (void)hipFree(x->d_all);
    if (x->h_all) (void)hipHostFree(x->h_all);
    delete x;
    return MALIO_OK;
  }
  if (x->kind == XCHG_SHM && x->base) munmap(x->base, x->bytes);
  if (x->owner) shm_unlink(x->name.c_str());
  delete x;
  return MALIO_OK;
}

}  // extern "C"
