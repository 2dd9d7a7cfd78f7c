// Host-side runtime of libgpk.so: device binding, streams, a caching device
// allocator (the optimiser calls gpk_fit hundreds of times: GP_regression_demo.py:44),
// the model handle and the blocked-Cholesky driver.
#pragma once
#include "gpk_internal.h"
#include <algorithm>
#include <functional>
#include <map>

struct StageTimer {
  struct Rec { hipEvent_t a, b; };
  std::vector<Rec> recs;
  double ms = 0.0;
  double count = 0.0;
  double work = 0.0;   // algorithmic flops (or bytes) of the recorded launches
  double work_pending = 0.0;
};

struct Runtime {
  bool inited = false;
  int device = 0;
  hipStream_t s_main = nullptr, s_panel = nullptr;
  hipStream_t s_pred = nullptr;   // y and prediction substitutions pipelined behind the factorisation (fused fit+predict)
  hipStream_t s_comm = nullptr;   // multi-rank processes: the panel broadcasts (off the panel chain); virtual ranks use s_pred
  hipStream_t s_rows = nullptr;   // tile pipeline of the chain-bound tail: the full-height panel GEMMs (s_panel keeps the leaves)
  bool cu_reserved = false;
  int ncu_reserved = 8;
  // resident leaf server (gpk_internal.h: LeafMailbox): its stream, its mailbox, the next sequence number, and the
  // chain stream of the open session (nullptr: none open, leaves are launched directly)
  hipStream_t s_leaf = nullptr;
  LeafMailbox* leaf_mb = nullptr;
  unsigned leaf_seq = 1;
  hipStream_t leaf_chain = nullptr;
  int leaf_server = 1;  // option: 0 = always launch the leaf kernel directly
  std::map<size_t, std::vector<void*>> pool;
  size_t pool_bytes = 0;
  // options
  int nb = 256;
  // single-GPU panel schedule: wide panels (K = nb_big) while the trailing matrix is large and the panel
  // chain hides behind the update, nb in the middle, nb_small at the tail where the chain IS the critical path
  int nb_big = 512, nb_big_until = 8192, nb_small = 128, nb_small_from = 4096, nb_auto = 1;
  // two-level blocking (single GPU, nb_auto): OUTER panels of nb_outer columns are applied to the trailing matrix as
  // ONE update with K = nb_outer (the kernel does 56 TFLOP/s at K = 2048 against 44 at K = 256: the C tile traffic
  // and the accumulator hand-over amortise over 8x the flops); inside an outer block column the panels above are the
  // inner level.  Active for Npad >= outer_from while more than outer_until columns remain; 0 = one level.
  int nb_outer = 2048, outer_from = 24576, outer_until = 8192;  // N = 16384: no gain; N = 32768: 230.6 -> 225.2 ms
  int lookahead = 1;
  int pred_gate_rows = 0;  // experiment: substitutions start when this few columns remain (0: at once)
  int ldk_pad = 0;      // extra doubles per row of K (single GPU)
  int dist_group = 2;   // panels per ownership block of a block-cyclic (multi-GPU) model
  // tile pipeline (potrf_tail) for the chain-bound last tail_rows columns (needs 128-wide panels there: nb_small_from);
  // 0 = the panel-level look-ahead everywhere
  int tail_rows = 4096;
  int tail_flags = 0;       // tile pipeline: flag-chained steps (leaf kernels on the chain stream, T kernels on the update stream, flags in HBM)
  unsigned* dflags = nullptr;  // 4 words per tile step (leaf flag, tt flag, two counters), zeroed at the start of a factorisation
  int tail_leaf_event = 0;  // tile pipeline: 1 = a separate event after every leaf (the full-height solve then starts ~20 us earlier)
  int tt_fused = 0;     // tile pipeline: the chain's two one-tile GEMMs of a step in ONE launch (k_gemm.hip: tt_fused_kernel); measured: no gain, off
  int tail_stream = 0;  // the tile pipeline's full-height GEMMs: 0 in front of the bulk on the update stream (default), 1 on a stream of their own (s_rows, a FOURTH stream)
  double last_potrf_enqueue_ms = 0.0, last_pipe_enqueue_ms = 0.0;  // host time spent queueing the last factorisation
  int timing = 0;
  int stamps = 0;  // diagnostic: device-side progress stamps per panel (gpk_get_option("stamp_panel_<p>" / "stamp_subst_<p>"), scripts/stamp_steps.py)
  unsigned long long* dstamps = nullptr;
  std::vector<unsigned long long> hstamps;
  int pred_small_below = 300;  // predict_update: same rule for the substitution GEMMs
  int small_tile_below = 256;  // panel_update: launches with fewer 128x128 tiles than this use the 64x64-tile kernel
  int test_gemm_k1 = 0;   // gpk_test_gemm_nt: split K at this column into two separately stored operands
  int pair_update = 1;    // two panels per trailing update (K doubled) for the far region: 0 never, 1 for Npad <= 8192, 2 always
  int pipe_interleave = 0;  // 1: host submits panel p's substitutions right after panel p (0: after the whole factorisation)
  int solve_gemm_above = 8;  // gpk_solve: more right-hand sides than this go through the blocked GEMM substitutions
  int fuse_predict = 1;  // bench / compute_prediction: overlap the prediction solve with the factorisation tail
  int xcd_sq = 1;              // XCD-aware 8x8-tile square order in the big GEMMs
  int xcd_sorted = 0;          // ... with the squares of a round sorted by fill (full squares first, diagonal squares last)
  int xcd_sq_min_tiles = 8192; // ... for launches with at least this many 128x128 tiles in the bounding rectangle
  int ncu = 256;         // CUs available to the GEMM streams
  int small_tile_rows = 4096;  // GEMMs with n == 128 use the 64x64 tile below this many rows... (see potrf)
  // comm
  int rank = 0, nranks = 1;
  void* nccl = nullptr;
  std::map<std::string, StageTimer> timers;
  std::vector<hipEvent_t> ev_free;
  std::vector<hipEvent_t> ev_dep_free;  // recycled dependency events (timing disabled) of destroyed models
};
extern Runtime G;

int rt_require();
int dev_alloc(void** p, size_t bytes);
void dev_free(void* p, size_t bytes);
template <class T> int dev_alloc_t(T** p, size_t n) { return dev_alloc((void**)p, n * sizeof(T)); }

// One factorisation's lease on the leaf server: the server is launched when the session opens, leaf_launch() on the
// session's chain stream posts tiles to it, and the destructor queues the quit message on that stream -- whichever way
// the factorisation's host code ends, the server is told to leave.  Nested sessions are no-ops.
struct LeafSession {
  bool owner = false;
  explicit LeafSession(hipStream_t chain);
  ~LeafSession();
  LeafSession(const LeafSession&) = delete;
  LeafSession& operator=(const LeafSession&) = delete;
};
int leaf_launch(const double* A, long lda, double* L, long ldl, double* Dinv, int* info, int col0, hipStream_t s);

// Scratch device buffers of one API call: returned to the pool when the call ends, whichever way it ends (the update
// stream is drained first, so nothing queued can still touch them).
struct Scratch {
  std::vector<std::pair<void*, size_t>> bufs;
  template <class T> int get(T** p, size_t n) {
    *p = nullptr;
    GPK_TRY(dev_alloc_t(p, n));
    bufs.emplace_back((void*)*p, n * sizeof(T));
    return 0;
  }
  ~Scratch() {
    if (!bufs.empty() && G.inited) (void)hipStreamSynchronize(G.s_main);
    for (auto& b : bufs) dev_free(b.first, b.second);
  }
};

void timer_begin(const char* stage, hipStream_t s, double work = 0.0);
void timer_end(const char* stage, hipStream_t s);
void timers_reset();
int timers_resolve();

struct gpk_model {
  int N = 0, d = 0, Npad = 0, nT = 0, NB = 0, nP = 0;
  // 1-D block-cyclic column ownership of K: panel g belongs to rank g % P; this rank
  // stores only its nloc block columns, side by side, in dK (Npad x ldk).  P = 1: ldk = Npad.
  int P = 1, r = 0, nloc = 0;
  long ldk = 0;
  // ownership blocks (P > 1): OB = dist_group consecutive panels belong to one rank (global block b -> rank b % P,
  // local block b / P), so that inside a group the next panel's owner already holds the panel it needs and no
  // broadcast sits on the panel chain; nloc = number of local ownership blocks, ldk = nloc * OB.  P == 1: OB = Npad.
  int OB = 0;
  int owner_of_col(long c) const { return P > 1 ? (int)((c / OB) % P) : 0; }
  long local_col(long c) const { return P > 1 ? (c / OB / P) * OB + c % OB : c; }   // c must be owned
  long local_from(long c) const {  // first local column whose global column is >= c (ldk if none)
    if (P == 1) return c;
    if (c >= Npad) return ldk;
    const long b = c / OB;
    if ((int)(b % P) == r) return (b / P) * OB + c % OB;
    long nb_ = b + 1;
    while ((int)(nb_ % P) != r) ++nb_;
    return std::min<long>(ldk, (nb_ / P) * OB);
  }
  bool virtual_rank = false;  // single-process emulation of rank r of P (tests): no RCCL calls
  // virtual ranks only: the P rank models of the emulated job, indexed by rank.  The "broadcast" / "all-gather" of a
  // virtual rank is a device copy from the peer's buffers; everything else is the code a real rank runs.
  gpk_model* const* vgroup = nullptr;
  // dependency events of the look-ahead schedules, owned by the model (P virtual ranks in one process must not
  // share them); see model_events()
  std::vector<hipEvent_t> ev;
  size_t dws_bytes = 0;       // size dws was allocated with (the nb_big option may change while the handle lives)
  double* shard_loc = nullptr;  // this rank's slice of a sharded prediction: [mean(Mpad) | var(Mpad)] (peers copy from it)
  // inner panels (the unit of the panel chain: leaf factorisations, substitutions): first column, width,
  // leading dimension of their storage, offset of element (c0, c0) in dL, outer panel they belong to
  std::vector<int> pc0, pw, pouter;
  std::vector<long> pld;
  std::vector<size_t> poff;
  // outer panels (the unit of the trailing update): first column, width, first / one-past-last inner panel; an outer
  // panel is stored packed, rows oc0.. x ow columns, ld = ow, at dL + ooff.  One level: outer o == inner panel o.
  int nO = 0;
  std::vector<int> oc0, ow, ofirst, oend;
  std::vector<size_t> ooff;
  double* Lo(int o) const { return dL + ooff[o]; }
  size_t L_elems = 0;
  double *dX = nullptr, *dy = nullptr, *dw = nullptr;
  double *dK = nullptr, *dKcopy = nullptr, *dL = nullptr, *dDinv = nullptr;
  double* dscal = nullptr;  // [0] sum log diag L  [1] w.w
  double* dws = nullptr;    // gemvT workspace 64*NB
  int* dinfo = nullptr;
  SEParams se;
  double sn2 = 0, jitter = 0;
  double nlml = 0, logdet = 0, quad = 0;
  int info = 0;
  bool has_y = false, keep_k = false;
  bool k_only = false;   // gpk_fit flag 2: K built and kept, nothing factored (classification factors B = I + W^1/2 K W^1/2 instead)
  bool factored = false;
  bool growable = false;  // gpk_fit flag 4: uniform 128-wide panels (a layout that does not depend on N), dX sized for Npad points:
                          // gpk_append extends the factor instead of refactoring
  size_t dX_elems = 0;
  double* Lpanel(int p) const { return dL + poff[p]; }
};

int model_create(gpk_model** out, int N, int d, bool keep_k, int P = -1, int r = -1, bool growable = false);
void model_destroy(gpk_model* m);
int decode_hyp(const double* log_hyp, int d, double jitter, gpk_model* m);
// prediction job fused into a fit: V = Kzx L^-T is pipelined panel by panel behind the factorisation
struct FusedPredict {
  const double* dZ; int M, Mpad;
  double *dR, *dV, *dmean, *dvar;
  double *h_mean = nullptr, *h_var = nullptr;  // host destinations: read back together with the scalars (one sync)
};
int fit_device(gpk_model* m, const FusedPredict* fp = nullptr);  // K build + potrf + w + scalars (inputs resident)
extern int g_potrf_lookahead_used;  // 1 if the last potrf_device recorded per-panel events in m->ev[0..nP)
extern int g_dist_lookahead_used;   // 1 if the last block-cyclic factorisation ran the look-ahead schedule
int potrf_device(gpk_model* m, const std::function<int(int)>* after_panel = nullptr);                                 // dK -> dL, dDinv
int model_events(gpk_model* m, int n);                         // at least n dependency events in m->ev
// Block-cyclic (P > 1) entry points work on the LOCAL rank models of one job: a real rank passes itself (n = 1,
// peers are other processes, exchange = RCCL), the single-process test harness passes all P virtual ranks
// (exchange = device copies).  Same schedule, same code.
int potrf_group(gpk_model* const* ms, int n);
int fit_group(gpk_model* const* ms, int n);
int predict_group(gpk_model* const* ms, int n, const double* dZ, int M, double* const* h_mean, double* const* h_var);
int build_K(gpk_model* m);                                      // dX -> dK (local block columns)
int factor_panel(gpk_model* m, int p, hipStream_t s);          // owner only
int panel_update(gpk_model* m, int p, int r0g, int c0g, int ncols, int lower, hipStream_t s);
int dist_update(gpk_model* m, int p, long gc_lo, long gc_hi, hipStream_t s);  // cyclic trailing update of the owned columns in [gc_lo, gc_hi)
int comm_bcast(double* buf, size_t count, int root, hipStream_t s);
int comm_allgather(const double* send, double* recv, size_t count, hipStream_t s);
// prediction for M test points resident at dZ: sharded over the ranks when P > 1 (each rank solves M/P
// right-hand sides, results all-gathered); dmv_out receives mean[0:M] at [0] and var[0:M] at [Mtot]
int predict_sharded(gpk_model* m, const double* dZ, int M, double* h_mean, double* h_var);
int nlml_grad_device(gpk_model* m, double* h_grad);
int potrf_forward(gpk_model* m, double* v);                   // dK -> dL, v <- L^-1 v (pipelined)
int rows_forward_solve(gpk_model* m, int Mpad, double* dR, double* dV);
int rows_backward_prepare(gpk_model* m, double* dU, double* dDinvT, hipStream_t s);  // U = L^T, transposed Dinv tiles
int rows_backward_solve(gpk_model* m, int Mpad, double* dR, double* dY, const double* dU, const double* dDinvT);
int trsv_forward(gpk_model* m, double* v);                      // v <- L^-1 v  (Npad vector)
int trsv_backward(gpk_model* m, double* v);                     // v <- L^-T v
int predict_device(gpk_model* m, const double* dZ, int M, int Mpad, double* dR, double* dV, double* dmean, double* dvar);
