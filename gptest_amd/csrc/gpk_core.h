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
  // Three streams, none of them CU-masked.  The update stream (normal priority) carries the bulk GEMMs; the panel chain
  // and the substitutions run above it.  The latency-critical leaf no longer needs reserved CUs: it takes the slot one
  // finished GEMM workgroup frees and the GEMM workgroups that share its CU pause while it runs (gpk_internal.h).
  hipStream_t s_main = nullptr, s_panel = nullptr;
  hipStream_t s_pred = nullptr;   // y and prediction substitutions pipelined behind the factorisation (fused fit+predict)
  hipStream_t s_comm = nullptr;   // multi-rank processes: the panel broadcasts (off the panel chain); virtual ranks use s_pred
  hipStream_t s_corner = nullptr; // the deferred corner updates of the lazy-corner schedule (created on first use: corner_stream_ensure)
  int s_corner_prio = -1;
  std::map<size_t, std::vector<void*>> pool;
  size_t pool_bytes = 0;
  // ---- options (gpk_set_option).  Panel schedule: nb = 0 is the automatic one -- 512-wide panels while more than
  // nb_big_until + 512 columns remain (K = 512: the update kernel runs faster and the longer panel chain still hides), 256 in
  // the middle, 128 for the last nb_small_from columns, where the chain IS the critical path; nb > 0: uniform panels.
  int nb = 0;
  int nb_big_until = 8192, nb_small_from = 4096;
  // two-level blocking (single GPU, automatic schedule): OUTER panels of nb_outer columns are applied to the trailing matrix
  // as ONE update with K = nb_outer; inside an outer block column the panels above are the inner level.  Active for
  // Npad >= 3 * outer_until while more than outer_until columns remain; nb_outer = 0: one level.
  // (N = 16384: no gain; N = 32768: 230.6 -> 225.2 ms.)
  int nb_outer = 2048, outer_until = 8192;
  int lookahead = 1;
  int dist_group = 2;   // panels (of the middle width) per ownership block of a block-cyclic (multi-GPU) model
  int dist_tail = 1;      // block-cyclic models: the last tail_rows columns as one tail block on one rank, factored there by the tile pipeline
  int tail_stream = 0;    // tile pipeline: the full-height solve and the look-ahead update of a step on a (high-priority) stream of their own (OFF: bit-identical, measured slower -- N=4096 2.79 -> 3.08-3.22 ms, 8192 7.78 -> 7.95, 16384 32.9 -> 33.2: profiles/r04_tail_stream_ab.txt)
  hipStream_t s_near = nullptr;
  int kbuild_split = 0;   // fused one-GPU fits: K's first block column as a launch of its own, so that panel 0 is factored beside the rest of the build (OFF: the build's waves fill every SIMD and panel 0 takes 0.58 instead of 0.41 ms beside it -- 20 us gained, none in the posterior)
  int se_scalar = 1;      // K / Kxz builds of models with d <= 8 in the scalar-operand form (SEScalarForm; 0: the LDS form, same bits)
  int dist_tail_alone = -1; // the tail's owner keeps out of the cyclic part (gpk_model::Pc): -1 where the work model says so, 0 never, 1 whenever P >= 3
  int dist_tail_chunk = 8; // ... whose panels are broadcast this many at a time
  int dist_pipeline = 1;  // multi-rank process: keep the substitution stream next to the communication stream and pipeline y / this rank's
                          // shard of the prediction behind the per-panel "available" events (fit_group_impl) -- if gpk_comm_init's probe finds the four
                          // streams running abreast (streams_abreast); 0: three streams, prediction afterwards; 2: four streams without the probe
  // tile pipeline (potrf_tail) for the chain-bound last tail_rows columns (needs 128-wide panels there: nb_small_from);
  // 0 = the panel-level look-ahead everywhere
  int tail_rows = 4096;
  int leaf_fat_rows = 6144;  // leaf_launch: the full-tile leaf once at most this many rows remain (option leaf_algo = 2)
  int pair_update = 1;    // two panels per trailing update (K doubled) for the far region: 0 never, 1 for Npad <= 8192, 2 always
  int fuse_predict = 1;   // bench / compute_prediction: overlap the prediction solve with the factorisation tail
  int tail_tile_kernel = -1; // (-1: automatic, for Npad < 12288) tile pipeline: the two one-tile GEMMs of a step on the LDS-free 64-wave kernel instead of the 64x64-tile one
  int pred_defer_from = -1;  // (-1: automatic, 5/8 of the columns from Npad = 12288 on; 0: off) pipelined prediction: panels that end at or before this column update only the columns below it right away; their updates of the columns from it on are queued, in panel order, when the substitution reaches it (0: off)
  // Lazy corner (potrf_device, one GPU, one-level schedule): with b the first panel boundary at or after corner_from, a panel that
  // ends before b applies its trailing update right away only to the trapezoid of columns < b; the corner [b:, b:] receives the
  // updates of those panels later, in panel order, as a few launches with K = corner_chunk columns (K-segment table of the 8-wave
  // kernel) on a low-priority stream of its own -- filler for the chain-bound part of the factorisation, at the rate the kernel
  // reaches at high K.  The last corner_last columns before b are applied together with the boundary panel's own update (so that
  // what the chain waits for at b is short).  Bit-identical to the immediate schedule.  -1: automatic, 0: off.
  int corner_from = 0, corner_chunk = 2048, corner_last = 256;
  int corner_from2 = 0, corner_from3 = 0;  // further boundaries (nested regions): the columns from corner_from2 on are left out of the first region's
                                           // catch-up and brought up to date when the chain reaches corner_from2 -- deferred work for the stretch between the two
  int corner_grid = 256;  // workgroups of a catch-up launch (persistent form of the 8-wave kernel: they share the tiles through a counter); 0: one
                          // workgroup per tile.  256 = one per CU: the other workgroup slot of every CU stays free for the panel chain
  unsigned* dctr = nullptr;  // tile counters of the persistent launches (64, used round-robin)
  unsigned dctr_next = 0;
  int corner_prio = 0;  // priority of the corner stream: 0 lowest, 1 that of the update stream, 2 highest
  int corner_hold = 0;  // no catch-up launch starts before the panel that begins at or after this column has been factored: the deferred work
                        // is held back until the factorisation turns chain-bound instead of competing with the bulk-bound first part
  // tile pipeline: leaf + the two one-tile GEMMs of a step as ONE launch (potf2_chain_kernel) that polls a flag for the update
  // stream's look-ahead update instead of waiting for an event in front of the launch.  0: off (default: measured slower, DESIGN.md
  // section 6), 1: on, 2: on, and the update stream's kernels poll the flags the chain kernel raises; -1: 2 where cross-stream flags
  // work (flags_ok, probed by gpk_init: they do not under a tool that serialises dispatches)
  int chain_fused = 0;
  int flags_ok = 0;
  unsigned flag_epoch = 0;     // one value per tile-pipeline run: a stale flag of an earlier run never satisfies a wait
  int small_tile_below = 256;  // panel_update: launches with fewer 128x128 tiles than this use the 64x64-tile kernel
  std::map<size_t, std::vector<double*>> hpin_pool;  // pinned read-back buffers of destroyed models, by size in doubles
  int timing = 0;
  int stamps = 0;  // diagnostic: device-side progress stamps per panel (gpk_get_option("stamp_panel_<p>" / "stamp_subst_<p>"), scripts/stamp_steps.py)
  unsigned long long* dstamps = nullptr;
  std::vector<unsigned long long> hstamps;
  double last_potrf_enqueue_ms = 0.0, last_pipe_enqueue_ms = 0.0;  // host time spent queueing the last factorisation
  // comm
  int rank = 0, nranks = 1;
  void* nccl = nullptr;
  std::map<std::string, StageTimer> timers;
  std::vector<hipEvent_t> ev_free;
  std::vector<hipEvent_t> ev_dep_free;  // recycled dependency events (timing disabled) of destroyed models
  // fixed parameters of the schedule (measured once; no longer options)
  static constexpr int nb_mid = 256, nb_big = 512, nb_small = 128;
  static constexpr int xcd_sq_min_tiles = 8192;  // XCD-aware 8x8-tile square order for launches with at least this many tiles
  static constexpr int solve_gemm_above = 8;     // gpk_solve: more right-hand sides than this go through the blocked GEMM substitutions
  int pred_small_below() const { return small_tile_below == 256 ? 300 : small_tile_below; }  // predict_update: the same rule for the substitution GEMMs
  bool nb_auto() const { return nb == 0; }
  int outer_from() const { return 3 * outer_until; }
};
extern Runtime G;

int rt_require();
int dev_alloc(void** p, size_t bytes);
void dev_free(void* p, size_t bytes);
template <class T> int dev_alloc_t(T** p, size_t n) { return dev_alloc((void**)p, n * sizeof(T)); }

// rows_left: rows of the matrix from this tile on (which of the two forms of the leaf: see gpk_core.hip)
int leaf_launch(const double* A, long lda, double* L, long ldl, double* Dinv, int* info, int col0, hipStream_t s, int rows_left = 1 << 30);

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
  // Tail block (round 4): from global column ct on (an ownership-block boundary; 0: none) every column belongs to ONE rank, t_own --
  // the chain-bound last columns then run through the one-GPU tile pipeline on that rank (no hand-over inside the tail is a
  // broadcast; its panels are broadcast behind the chain).  Local layout of a rank: its nloc_c cyclic blocks of the columns
  // below ct side by side, then (on t_own only) the Npad - ct tail columns.
  int ct = 0, t_own = 0, nloc_c = 0, p_tail = 0;  // p_tail: the panel that starts at ct
  // Pc: the ranks that share the cyclic part.  P, or -- where the tail block alone is at least a fair share of the work (model_create:
  // P >= 3 at the default sizes) -- P - 1: the tail's owner is then rank P - 1 and keeps out of the cyclic part (its peers' map is
  // the cyclic one over Pc ranks), instead of carrying ALL updates of the tail block on top of a full cyclic share.
  int Pc = 1;
  int owner_of_col(long c) const {
    if (ct > 0 && c >= ct) return t_own;
    return P > 1 ? (int)((c / OB) % Pc) : 0;
  }
  long local_col(long c) const {  // c must be owned
    if (ct > 0 && c >= ct) return (long)nloc_c * OB + (c - ct);
    return P > 1 ? (c / OB / Pc) * OB + c % OB : c;
  }
  long local_from(long c) const {  // first local column whose global column is >= c (ldk if none)
    if (P == 1 && ct == 0) return c;
    if (c >= Npad) return ldk;
    if (ct > 0 && c >= ct) return r == t_own ? (long)nloc_c * OB + (c - ct) : ldk;
    const long nb_all = (ct > 0 ? ct : Npad) / OB;  // cyclic blocks
    const long b = c / OB;
    if (r < Pc) {
      if ((int)(b % Pc) == r) return (b / Pc) * OB + c % OB;
      long nb_ = b + 1;
      while (nb_ < nb_all && (int)(nb_ % Pc) != r) ++nb_;
      if (nb_ < nb_all) return (nb_ / Pc) * OB;
    }
    return (ct > 0 && r == t_own) ? (long)nloc_c * OB : ldk;  // no cyclic block left: the tail block (its owner) or nothing
  }
  // the global column ranges this rank stores, in local order: (first global column, columns, first local column)
  struct ColRange { int gc0, ncols; long lc0; };
  std::vector<ColRange> owned_ranges() const {
    std::vector<ColRange> v;
    if (P == 1 && ct == 0) { v.push_back({0, Npad, 0}); return v; }
    const int nb_all = (ct > 0 ? ct : Npad) / OB;
    int lb = 0;
    for (int b = 0; b < nb_all && r < Pc; ++b)
      if (b % Pc == r) { v.push_back({b * OB, OB, (long)lb * OB}); ++lb; }
    if (ct > 0 && r == t_own) v.push_back({ct, Npad - ct, (long)nloc_c * OB});
    return v;
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
  // scalar-operand form of the K builder (SEScalarForm): the scaled training points + norms (left there by the K build; every
  // later Kxz build of this model reads them as its column points) and, grown on demand, the scaled test points of a Kxz build
  hipEvent_t ev_kfirst = nullptr;  // (set by fit_enqueue around its potrf_device call) K's first block column is built; the rest follows on s_main
  double *dXs = nullptr, *dZs = nullptr;
  size_t dXs_elems = 0, dZs_elems = 0;
  SEScalarForm k_form(int rows_ready = 0) const;            // for a K build (rows = columns = the training points)
  int kxz_form(int Mpad, SEScalarForm* out);                // for a Kxz build of Mpad test points (after the K build)
  double *dK = nullptr, *dKcopy = nullptr, *dL = nullptr, *dDinv = nullptr;
  double* dscal = nullptr;  // [0] sum log diag L  [1] w.w
  double* hpin = nullptr;   // pinned host staging of a fit's read-back: [sum log diag L, w.w, info, -, mean(M), var(M)]
  size_t hpin_doubles = 0;
  double* dws = nullptr;    // gemvT workspace 64*NB
  int* dinfo = nullptr;
  unsigned* dflags = nullptr;  // three flags per tile column p (fused chain step): [p] the update stream has applied the look-ahead update of step p,
                               // [nT + p] the chain kernel has written L(p,p) and Dinv(p), [2 nT + p] ... and L(p+1,p)
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

// *info of a factorisation read back on the host: > 0 is the first non-positive pivot (LAPACK style), GPK_INFO_TIMEOUT and above
// means that a bounded device-side wait ran out (the fused chain step: k_potf2.hip) -- an error (-3), never a pivot.
constexpr int GPK_INFO_TIMEOUT = 1 << 30;
static inline int info_timeout_rc(int hinfo) {
  return hinfo >= GPK_INFO_TIMEOUT ? gpk_set_err(-3, "a device-side wait of the factorisation timed out (results discarded)") : 0;
}
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
int corner_stream_ensure();                                     // the corner stream exists (with the priority option corner_prio asks for)
int corner_boundary(const gpk_model* m);                        // boundary column of the lazy-corner schedule for this model (0: off)
extern int g_potrf_corner[3];                                   // boundaries the last factorisation deferred its far columns from (0: none)
// Block-cyclic (P > 1) entry points work on the LOCAL rank models of one job: a real rank passes itself (n = 1,
// peers are other processes, exchange = RCCL), the single-process test harness passes all P virtual ranks
// (exchange = device copies).  Same schedule, same code.
int potrf_group(gpk_model* const* ms, int n);
int fit_group(gpk_model* const* ms, int n);
int predict_group(gpk_model* const* ms, int n, const double* dZ, int M, double* const* h_mean, double* const* h_var);
// fit + sharded prediction in one go: the y and prediction substitutions pipelined behind the per-panel events of the block-cyclic factorisation
int fit_predict_group(gpk_model* const* ms, int n, const double* dZ, int M, double* const* h_mean, double* const* h_var);
extern int g_dist_tail_from;        // first panel the last block-cyclic factorisation ran through the tile pipeline on the tail's owner (-1: none)
extern int g_dist_streams_abreast;  // gpk_comm_init's probe of the four-stream layout: 1 abreast, 0 time-sliced (three streams kept), -1 not probed
extern double g_dist_streams_worst_us;
int streams_abreast(hipStream_t* st, int n, int reps, double* worst_us, double* best_us);  // gpk_comm.hip
extern int g_potrf_tail_stream;     // 1 if the last tile pipeline used the near stream
extern int g_kbuild_split;          // 1 if the last fused fit built K in two launches (first block column, rest)
extern int g_dist_tail_owner;       // ... its owner; P + owner if that rank keeps out of the cyclic part
extern int g_dist_tail_col;         // ... and its first column
extern int g_dist_pipelined;        // 1 if the last block-cyclic fit queued its substitutions behind the panel events
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
int model_unpack_L(gpk_model* m, double* L_out_host);                                 // the packed factor as a dense N x N lower matrix on the host
int launch_pad_identity(double* K, long ld, int N, int Npad, hipStream_t s);          // identity in the padding rows / columns of a caller matrix
