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
#include <vector>
#include "../../include/malio.h"

struct malio_xchg {
  int rank = 0, world = 0, row = 0;
  bool owner = false;
  size_t bytes = 0;
  char *base = nullptr;
  uint64_t epoch = 0;
  std::string name;
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

int malio_xchg_all_gather(malio_xchg_t x, const double *in, double *out_all, double timeout_s) {
  if (!x || !in || !out_all) return MALIO_ERR_BAD_ARG;
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

int malio_xchg_reduce(malio_xchg_t x, const double *row_in, int ns, const double *guess4, double *sums_out,
                      double *extrema4_out, double timeout_s) {
  if (!x || !row_in || !sums_out || !extrema4_out || ns < 0 || ns + 4 > x->row) return MALIO_ERR_BAD_ARG;
  x->all.resize((size_t)x->world * x->row);
  int rc = malio_xchg_all_gather(x, row_in, x->all.data(), timeout_s);
  if (rc != MALIO_OK) return rc;
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
  if (x->base) munmap(x->base, x->bytes);
  if (x->owner) shm_unlink(x->name.c_str());
  delete x;
  return MALIO_OK;
}

}  // extern "C"
