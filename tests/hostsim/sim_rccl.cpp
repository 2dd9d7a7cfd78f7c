// TEST INFRASTRUCTURE ONLY (tests/hostsim): inter-process stand-in for RCCL, so that the REAL-RANK branches of libgpk's
// host code (comm_bcast / comm_allgather / gpk_comm_allreduce_max, fit_group / predict_group with n = 1, the multi-rank
// leg of gpk_bench_posterior) run with 2, 4, ... real processes on the CPU.  A communicator is a shared-memory segment
// (/dev/shm/gpk_hostsim_<id>): a header with a counting barrier and the per-rank call descriptors, then one data slot per
// rank.  Collectives are blocking and chunked through the slots; the call descriptors (operation, count, root, sequence
// number) of all ranks are compared inside every collective, which is how a schedule that lets ranks disagree on the
// order or shape of the exchanges shows up as an error instead of silent corruption.
#include <rccl/rccl.h>

#include <atomic>
#include <cerrno>
#include <cstdio>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

namespace {
constexpr int kMaxRanks = 16;
constexpr size_t kSlot = 1u << 20;  // bytes per rank and chunk
struct Call { int op; int root; unsigned long long count; unsigned long long seq; };
struct Shared {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> attached;
  std::atomic<int> failed;  // a rank that hit an error leaves it here: nobody waits for it any longer
  Call call[kMaxRanks];
  char pad[4096 - 4 * sizeof(std::atomic<int>) - kMaxRanks * sizeof(Call)];
};
static_assert(sizeof(Shared) == 4096, "header layout");
double timeout_s() {
  const char* e = getenv("HOSTSIM_RCCL_TIMEOUT_S");
  return e ? atof(e) : 120.0;
}
double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
}  // namespace

struct hostsim_comm {
  int n = 1, rank = 0;
  Shared* sh = nullptr;
  char* data = nullptr;
  size_t bytes = 0;
  unsigned long long seq = 0;
  char name[64] = {0};
};

static ncclResult_t barrier(hostsim_comm* c) {
  Shared* sh = c->sh;
  const int gen = sh->generation.load(std::memory_order_acquire);
  if (sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->n) {
    sh->arrived.store(0, std::memory_order_relaxed);
    sh->generation.fetch_add(1, std::memory_order_acq_rel);
    return ncclSuccess;
  }
  const double t0 = now_s(), limit = timeout_s();
  long spins = 0;
  while (sh->generation.load(std::memory_order_acquire) == gen) {
    if (sh->failed.load(std::memory_order_relaxed)) return ncclSystemError;
    if ((++spins & 255) == 0) {
      if (now_s() - t0 > limit) {
        sh->failed.store(1);
        std::fprintf(stderr, "hostsim rccl: rank %d waited %.0f s in collective #%llu -- the other ranks never arrived\n", c->rank, limit, c->seq);
        return ncclSystemError;
      }
      usleep(50);
    } else {
      sched_yield();
    }
  }
  return ncclSuccess;
}

// every rank announces what it is about to do; after the barrier all descriptors must agree
static ncclResult_t match(hostsim_comm* c, int op, int root, size_t count) {
  c->seq++;
  c->sh->call[c->rank] = Call{op, root, (unsigned long long)count, c->seq};
  if (barrier(c) != ncclSuccess) return ncclSystemError;
  for (int r = 0; r < c->n; ++r) {
    const Call& o = c->sh->call[r];
    if (o.op != op || o.root != root || o.count != count || o.seq != c->seq) {
      std::fprintf(stderr, "hostsim rccl: collective #%llu mismatch: rank %d does op %d root %d count %zu, rank %d does op %d root %d count %llu (its #%llu)\n",
                   c->seq, c->rank, op, root, count, r, o.op, o.root, o.count, o.seq);
      c->sh->failed.store(1);
      return ncclInvalidUsage;
    }
  }
  return barrier(c);  // nobody overwrites its descriptor before everyone has compared
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "success";
    case ncclSystemError: return "hostsim rccl: a rank failed or never arrived (see stderr)";
    case ncclInvalidUsage: return "hostsim rccl: ranks disagree on a collective (see stderr)";
    default: return "hostsim rccl: invalid argument";
  }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof *id);
  static int counter = 0;
  std::snprintf(id->internal, sizeof id->internal, "/gpk_hostsim_%d_%d_%llx", (int)getpid(), counter++, (unsigned long long)(now_s() * 1e6));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int n, ncclUniqueId id, int rank) {
  if (n < 1 || n > kMaxRanks || rank < 0 || rank >= n) return ncclInvalidArgument;
  hostsim_comm* c = new hostsim_comm();
  c->n = n;
  c->rank = rank;
  if (n == 1) { *out = c; return ncclSuccess; }
  id.internal[sizeof id.internal - 1] = 0;
  if (id.internal[0] != '/') { delete c; return ncclInvalidArgument; }
  std::snprintf(c->name, sizeof c->name, "%s", id.internal);
  c->bytes = sizeof(Shared) + (size_t)n * kSlot;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);  // a fresh segment is zero-filled: a valid initial barrier state
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
    std::fprintf(stderr, "hostsim rccl: shm_open/ftruncate(%s) failed: %s\n", c->name, std::strerror(errno));
    if (fd >= 0) close(fd);
    delete c;
    return ncclSystemError;
  }
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return ncclSystemError; }
  c->sh = static_cast<Shared*>(p);
  c->data = static_cast<char*>(p) + sizeof(Shared);
  c->sh->attached.fetch_add(1);
  if (barrier(c) != ncclSuccess) {  // as the real call: returns once every rank has joined
    munmap(p, c->bytes);
    shm_unlink(c->name);
    delete c;
    return ncclSystemError;
  }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->sh) {
    const bool last = c->sh->attached.fetch_sub(1) == 1;
    munmap(c->sh, c->bytes);
    if (last) shm_unlink(c->name);
  }
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* s, void* r, size_t n, ncclDataType_t, int root, ncclComm_t c, hipStream_t) {
  if (!c || root < 0 || root >= c->n) return ncclInvalidArgument;
  if (c->n == 1) { if (s != r) std::memmove(r, s, n * 8); return ncclSuccess; }
  if (ncclResult_t e = match(c, 1, root, n)) return e;
  const size_t bytes = n * 8;
  for (size_t off = 0; off < bytes; off += kSlot) {
    const size_t len = bytes - off < kSlot ? bytes - off : kSlot;
    if (c->rank == root) std::memcpy(c->data, static_cast<const char*>(s) + off, len);
    if (ncclResult_t e = barrier(c)) return e;
    if (c->rank != root) std::memcpy(static_cast<char*>(r) + off, c->data, len);
    else if (s != r) std::memmove(static_cast<char*>(r) + off, static_cast<const char*>(s) + off, len);
    if (ncclResult_t e = barrier(c)) return e;
  }
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* s, void* r, size_t n, ncclDataType_t, ncclComm_t c, hipStream_t) {
  if (!c) return ncclInvalidArgument;
  if (c->n == 1) { std::memmove(r, s, n * 8); return ncclSuccess; }
  if (ncclResult_t e = match(c, 2, -1, n)) return e;
  const size_t bytes = n * 8;
  for (size_t off = 0; off < bytes; off += kSlot) {
    const size_t len = bytes - off < kSlot ? bytes - off : kSlot;
    std::memcpy(c->data + (size_t)c->rank * kSlot, static_cast<const char*>(s) + off, len);
    if (ncclResult_t e = barrier(c)) return e;
    for (int q = 0; q < c->n; ++q) std::memcpy(static_cast<char*>(r) + (size_t)q * bytes + off, c->data + (size_t)q * kSlot, len);
    if (ncclResult_t e = barrier(c)) return e;
  }
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* s, void* r, size_t n, ncclDataType_t, ncclRedOp_t, ncclComm_t c, hipStream_t) {
  if (!c) return ncclInvalidArgument;
  if (c->n == 1) { if (s != r) std::memmove(r, s, n * 8); return ncclSuccess; }
  if (ncclResult_t e = match(c, 3, -1, n)) return e;
  const size_t per = kSlot / 8;
  for (size_t off = 0; off < n; off += per) {
    const size_t len = n - off < per ? n - off : per;
    std::memcpy(c->data + (size_t)c->rank * kSlot, static_cast<const double*>(s) + off, len * 8);
    if (ncclResult_t e = barrier(c)) return e;
    double* out = static_cast<double*>(r) + off;
    for (size_t i = 0; i < len; ++i) {
      double m = reinterpret_cast<const double*>(c->data)[i];
      for (int q = 1; q < c->n; ++q) {
        const double v = reinterpret_cast<const double*>(c->data + (size_t)q * kSlot)[i];
        if (v > m) m = v;
      }
      out[i] = m;
    }
    if (ncclResult_t e = barrier(c)) return e;
  }
  return ncclSuccess;
}
