// RCCL plumbing: one process per GPU, the communicator lives inside libgpk.so.
// The host (bench.py / gptest_amd.dist) only ferries the 128-byte unique id.
// The reference has no distributed code at all; this is the exchange step that
// BASELINE.json's north_star prescribes (panel broadcast over xGMI).
#include "../../include/gpk.h"
#include "gpk_core.h"
#include <rccl/rccl.h>
#include <cstring>
#include <vector>

#define GPK_NCCL(x)                                                                                          \
  do {                                                                                                       \
    ncclResult_t r_ = (x);                                                                                   \
    if (r_ != ncclSuccess)                                                                                   \
      return gpk_set_err(-3, "%s failed: %s (%s:%d)", #x, ncclGetErrorString(r_), __FILE__, __LINE__);      \
  } while (0)

static_assert(GPK_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

int comm_bcast(double* buf, size_t count, int root, hipStream_t s) {
  if (G.nranks <= 1) return 0;
  GPK_NCCL(ncclBroadcast(buf, buf, count, ncclDouble, root, (ncclComm_t)G.nccl, s));
  return 0;
}

// every rank contributes `count` doubles; recv holds nranks*count
int comm_allgather(const double* send, double* recv, size_t count, hipStream_t s) {
  if (G.nranks <= 1) return 0;
  GPK_NCCL(ncclAllGather(send, recv, count, ncclDouble, (ncclComm_t)G.nccl, s));
  return 0;
}

extern "C" {

int gpk_comm_unique_id(void* out128) {
  if (!out128) return gpk_set_err(-1, "gpk_comm_unique_id: null pointer");
  ncclUniqueId id;
  GPK_NCCL(ncclGetUniqueId(&id));
  std::memcpy(out128, &id, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

}  // extern "C"

// Do these streams run ABREAST?  `reps` rounds of one spinning wave (10 us) on each, queued round-robin; per stream the time from
// its first to its last launch.  On this runtime three streams always do; a fourth may land on a hardware queue that is time-sliced
// with another stream's on one pipe -- whether it does differs from box to box and run to run (profiles/r04_stream_probe*.txt,
// r04_batched_pipeline_again.txt) -- and then every small launch on the two sharing streams takes 40-70 us instead of 11.
// worst_us: the slowest stream's time per launch.  Costs reps x 10 us.
int streams_abreast(hipStream_t* st, int n, int reps, double* worst_us, double* best_us) {
  std::vector<hipEvent_t> e0(n), e1(n);
  for (int i = 0; i < n; ++i) { GPK_HIP(hipEventCreate(&e0[i])); GPK_HIP(hipEventCreate(&e1[i])); }
  for (int i = 0; i < n; ++i) GPK_TRY(launch_spin(100, st[i]));  // the queues exist
  for (int i = 0; i < n; ++i) GPK_HIP(hipStreamSynchronize(st[i]));
  for (int i = 0; i < n; ++i) GPK_HIP(hipEventRecord(e0[i], st[i]));
  for (int r = 0; r < reps; ++r)
    for (int i = 0; i < n; ++i) GPK_TRY(launch_spin(1000, st[i]));
  for (int i = 0; i < n; ++i) GPK_HIP(hipEventRecord(e1[i], st[i]));
  for (int i = 0; i < n; ++i) GPK_HIP(hipStreamSynchronize(st[i]));
  double worst = 0.0, best = 1e30;
  for (int i = 0; i < n; ++i) {
    float ms = 0;
    GPK_HIP(hipEventElapsedTime(&ms, e0[i], e1[i]));
    worst = std::max(worst, 1e3 * ms / reps);
    best = std::min(best, 1e3 * ms / reps);
    (void)hipEventDestroy(e0[i]);
    (void)hipEventDestroy(e1[i]);
  }
  *worst_us = worst;
  if (best_us) *best_us = best;
  return 0;
}
static constexpr double GPK_ABREAST_MAX_US = 16.0;
int g_dist_streams_abreast = -1;   // what gpk_comm_init's probe of the four-stream layout found: 1 abreast, 0 time-sliced (substitution stream given up), -1 not probed
double g_dist_streams_worst_us = 0.0;

extern "C" {

int gpk_comm_init(int rank, int nranks, const void* uid128) {
  GPK_TRY(rt_require());
  if (G.nccl) return gpk_set_err(-1, "gpk_comm_init: communicator already initialised");
  if (nranks < 1 || rank < 0 || rank >= nranks) return gpk_set_err(-1, "gpk_comm_init: bad rank %d / %d", rank, nranks);
  if (nranks == 1) {
    G.rank = 0;
    G.nranks = 1;
    return 0;
  }
  if (!uid128) return gpk_set_err(-1, "gpk_comm_init: null unique id");
  ncclUniqueId id;
  std::memcpy(&id, uid128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c;
  GPK_NCCL(ncclCommInitRank(&c, nranks, id, rank));
  G.nccl = c;
  G.rank = rank;
  G.nranks = nranks;
  // A multi-rank process gets a COMMUNICATION stream on top of the three of the one-GPU layout: the panel broadcasts run there, off
  // the panel chain (potrf_group).  The substitution stream stays (round 4): y and this rank's shard of the prediction are
  // substituted on it behind the per-panel "available" events (fit_group_impl), as on one GPU.
  // (option dist_pipeline = 0: the round-3 layout -- three streams, the substitution stream given up for the communication stream,
  //  prediction after the factorisation.  A fourth stream is not free on this runtime: two hardware queues can end up time-sliced on one
  //  pipe, profiles/r04_stream_probe.txt; which layout wins on a real node has to be measured there.)
  if (!G.dist_pipeline && G.s_pred && G.s_pred != G.s_panel) {
    (void)hipStreamSynchronize(G.s_pred);
    (void)hipStreamDestroy(G.s_pred);
    G.s_pred = nullptr;
  }
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (hipStreamCreateWithPriority(&G.s_comm, hipStreamNonBlocking, hi) != hipSuccess) {
    // the communicator is up: go on without a stream of its own (the broadcasts then run on the panel chain's stream,
    // comm_stream() in gpk_core.hip) rather than leave the process half initialised
    (void)hipGetLastError();
    G.s_comm = nullptr;
  }
  // dist_pipeline = 1 (default): keep the four-stream layout only if the four DO run abreast in this process on this GPU -- a
  // time-sliced pair costs every small launch of the chain 4x (N = 4096 one-GPU with a fourth stream: 2.85 -> 6.9-9.0 ms on a box
  // that draws the short straw); else the three-stream layout of dist_pipeline = 0.  dist_pipeline = 2: four streams, no probe.
  g_dist_streams_abreast = -1;
  if (G.dist_pipeline == 1 && G.s_comm && G.s_pred && G.s_pred != G.s_panel) {
    hipStream_t st[4] = {G.s_main, G.s_panel, G.s_pred, G.s_comm};
    double worst = 0.0, best = 0.0;
    if (streams_abreast(st, 4, 40, &worst, &best) == 0) {
      g_dist_streams_worst_us = worst;
      g_dist_streams_abreast = (worst <= GPK_ABREAST_MAX_US) ? 1 : 0;  // (the launches spin for exactly 10 us; 11-12 abreast, 19-70 time-sliced)
      if (!g_dist_streams_abreast) {
        (void)hipStreamSynchronize(G.s_pred);
        (void)hipStreamDestroy(G.s_pred);
        G.s_pred = nullptr;
      }
    } else {
      (void)hipGetLastError();
    }
  }
  return 0;
}

int gpk_comm_rank(void) { return G.rank; }
int gpk_comm_size(void) { return G.nranks; }

int gpk_comm_allreduce_max(double* inout, int n) {
  GPK_TRY(rt_require());
  if (G.nranks <= 1 || n <= 0) return 0;
  double* d = nullptr;
  GPK_TRY(dev_alloc_t(&d, (size_t)n));
  hipStream_t s = G.s_main;
  int rc = 0;
  do {
    if (hipMemcpyAsync(d, inout, (size_t)n * 8, hipMemcpyHostToDevice, s) != hipSuccess) { rc = gpk_set_err(-2, "H2D failed"); break; }
    ncclResult_t r = ncclAllReduce(d, d, n, ncclDouble, ncclMax, (ncclComm_t)G.nccl, s);
    if (r != ncclSuccess) { rc = gpk_set_err(-3, "ncclAllReduce failed: %s", ncclGetErrorString(r)); break; }
    if (hipMemcpyAsync(inout, d, (size_t)n * 8, hipMemcpyDeviceToHost, s) != hipSuccess) { rc = gpk_set_err(-2, "D2H failed"); break; }
    if (hipStreamSynchronize(s) != hipSuccess) { rc = gpk_set_err(-2, "allreduce sync failed"); break; }
  } while (0);
  dev_free(d, (size_t)n * 8);
  return rc;
}

int gpk_comm_barrier(void) {
  double x = 0.0;
  GPK_TRY(gpk_comm_allreduce_max(&x, 1));
  if (G.inited) GPK_HIP(hipDeviceSynchronize());
  return 0;
}

int gpk_comm_destroy(void) {
  if (G.nccl) {
    (void)hipDeviceSynchronize();
    ncclCommDestroy((ncclComm_t)G.nccl);
    G.nccl = nullptr;
  }
  G.rank = 0;
  G.nranks = 1;
  // back to the single-GPU stream layout: the communication stream goes, the substitution stream a multi-rank process gave
  // up for it (gpk_comm_init) comes back, so that a process that once joined a communicator runs the pipelined one-GPU
  // path afterwards
  if (G.inited) {
    if (G.s_comm) {
      (void)hipStreamSynchronize(G.s_comm);
      (void)hipStreamDestroy(G.s_comm);
      G.s_comm = nullptr;
    }
    if (!G.s_pred) {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      if (hipStreamCreateWithPriority(&G.s_pred, hipStreamNonBlocking, hi) != hipSuccess) {
        (void)hipGetLastError();
        G.s_pred = nullptr;
      }
    }
  }
  return 0;
}

}  // extern "C"
