// Internal declarations shared by the HIP translation units of libgpk.so.
// gfx950 (MI355X / CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define GPK_T 128  // tile edge: every padded dimension is a multiple of this

int gpk_set_err(int code, const char* fmt, ...);

#define GPK_HIP(x)                                                                             \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess)                                                                      \
      return gpk_set_err(-2, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define GPK_TRY(x)          \
  do {                      \
    int r_ = (x);           \
    if (r_ != 0) return r_; \
  } while (0)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---------------------------------------------------------------- GEMM (NT)
// C[m x n] (op)= A[m x k] * B[n x k]^T, all row-major fp64, MFMA 16x16x4 f64.
// m, n multiples of the block tile (128 or 64), k multiple of 16.
// "Polite co-residency" (no CU is reserved for the latency-critical leaf any more): a table of guest counters, one per
// physical CU (index = XCC_ID * 256 + HW_ID bits [15:8]: se, sh, cu), in device memory.  The leaf raises its CU's counter
// while it runs; wave 0 of every 8-wave GEMM workgroup reads its CU's counter once per K-slab (a load that rides on the
// slab's own vmcnt wait) and, while it is non-zero, sleeps -- the workgroup then stops at its next barrier -- so the leaf has
// the CU's fp64 pipe to itself although it only ever got the slot one finished GEMM workgroup left.  Bounded: a GEMM wave
// sleeps at most GPK_GUEST_LIMIT_TICKS of the 100 MHz clock per visit.
#define GPK_GUEST_SLOTS 2048
constexpr unsigned long long GPK_GUEST_LIMIT_TICKS = 30000ull;  // 300 us
extern unsigned* g_cu_guest;  // nullptr: nothing polls, nothing announces (k_gemm.hip)
extern hipStream_t g_chain_stream;  // the panel-chain stream (gpk_init): 64x64-tile GEMMs queued on it run at raised wave priority

struct GemmNT {
  const double* A; long lda;
  const double* B; long ldb;
  double* C; long ldc;
  int m, n, k;
  int sub;    // 0: C = A*B^T      1: C = C - A*B^T
  int lower;  // 1: compute only tiles that touch the lower triangle (global row >= global col)
  // block-cyclic column map (multi-GPU trailing update). cyc_P == 0 -> identity map.  C column 0 is local column
  // cyc_lg0 * cyc_NB + cyc_c0 of this rank; local ownership block b holds global columns ((b * P + r) * NB ...).
  int cyc_P, cyc_r, cyc_NB, cyc_lg0, cyc_c0;
  long row_g0;   // global row index of C row 0
  long brow_g0;  // global row index of B row 0 (B tile row = gcol(tj) - brow_g0 when cyc_P > 0 or lower)
  int xcd_sq;    // 8-wave kernel: XCD-aware 8x8-tile square order (single GPU)
  // optional second K segment (8-wave kernel only): columns [k1, k) of the operands come from A2/B2 (two panels
  // stored separately applied as ONE update: C -= [A A2][B B2]^T).  k1 == 0: single segment.
  const double* A2; long lda2;
  const double* B2; long ldb2;
  int k1;
  const unsigned* guest;  // the guest table (filled in by launch_gemm_nt; nullptr = do not poll)
  int poll_mask;  // 8-wave kernel: the guest counter is requested in the slabs kt with (kt & poll_mask) == poll_mask (0: every slab)
  int prio;  // 64x64-tile kernel: raise the waves' issue priority (s_setprio): set by launch_gemm_nt for launches on the panel-chain stream
  unsigned long long* trace;  // timing builds only (make DBG=1): per-workgroup time stamps
  // 64x64-tile kernel: every workgroup waits (bounded; a time-out poisons *wait_info) until *wait_flag has reached wait_val before
  // it touches memory -- the launch can be resident and ready BEFORE its producer on another stream has finished (tile pipeline)
  const unsigned* wait_flag; unsigned wait_val; int* wait_info;
};
// K-segment table of the 8-wave kernel: the K range of ONE update is the concatenation of up to GPK_MAXSEG separately stored
// operand pairs (factored panels, each packed with its own leading dimension), applied in table order inside one MFMA chain
// per output element -- bit for bit the result of one launch per segment in that order, with one C load / store instead
// of one per segment (the deferred corner update of the factorisation, potrf_device; the paired trailing update).
#define GPK_MAXSEG 32
struct GemmSeg {
  const double* A;  // rows of C tile row 0 (the kernel adds ti * 128 * lda)
  const double* B;  // rows of C tile column 0
  int lda, ldb;
  int k;            // columns of this segment (multiple of 16)
  int pad_;
};
struct GemmSegs {
  int n;
  int pad_;
  GemmSeg s[GPK_MAXSEG];
};
// tile128 = true: the 8-wave 128x128-tile kernel (LDS-DMA staged); false: the 64x64-tile kernel (4 waves, 2x2 MFMA tiles each).
int launch_gemm_nt(const GemmNT& g, bool tile128, hipStream_t s);
// the 8-wave kernel over a K-segment table: g.A / g.B / g.k / g.k1 are ignored (k = sum of the segments), single GPU only
// persist_grid > 0 (and a device counter): the persistent form -- that many workgroups share the tiles through the counter
int launch_gemm_nt_segs(const GemmNT& g, const GemmSegs& segs, hipStream_t s, int persist_grid = 0, unsigned* counter = nullptr);
// one 128 x 128 x 128 product on 64 waves without LDS (the chain's one-tile GEMMs); bitwise the other kernels
int launch_gemm_tile128(const GemmNT& g, hipStream_t s);
// the 8-wave kernel, with the XCD-aware tile order for the largest updates
int launch_gemm_auto(const GemmNT& g, hipStream_t s);
// ---------------------------------------------------------------- leaf factor
// Factor the 128x128 diagonal tile A (lda) -> L tile (ldl, strict upper zeroed) and
// Dinv = inv(L) (dense 128x128, ld 128, strict upper zeroed).  *info gets
// (col0 + j + 1) of the first non-positive pivot (only if it is still 0).
// fat: the caller expects an EMPTY CU to be easy to find (little else queued): the 1024-thread / 132 KiB form, which shares
// its CU with nobody, instead of the slot-sized one (k_potf2.hip).  Same bits either way.
int launch_potf2_trtri_128(const double* A, long lda, double* L, long ldl, double* Dinv, int* info,
                           int col0, int n_valid, hipStream_t s, int fat = 0);

// One launch per chain step of the tile pipeline (k_potf2.hip: potf2_chain_kernel): the full-tile leaf of tile (p,p) at A, then
// L(p+1,p) = A(p+1,p) Dinv^T (written to L + 128 * ldl) and A(p+1,p+1) -= L(p+1,p) L(p+1,p)^T in place, bit for bit what the leaf and
// two one-tile GEMMs produce.  flag != nullptr: rows p+1 are not touched before *flag has reached `want` (set by launch_flag_set
// on the stream that updates them); the wait is bounded and poisons *info with GPK_INFO_TIMEOUT when it runs out.
// done_leaf / done_t1 (optional): flags the kernel raises to `want` once L(p,p) and Dinv(p), and then L(p+1,p), are in memory --
// what the update stream's full-height solve and look-ahead update of the same step poll (GemmNT::wait_flag) instead of
// waiting for this kernel to end.
int launch_potf2_chain(double* A, long lda, double* L, long ldl, double* Dinv, int* info, int col0, const unsigned* flag, unsigned want,
                       hipStream_t s, unsigned* done_leaf = nullptr, unsigned* done_t1 = nullptr, int do_t2 = 1);
int launch_flag_set(unsigned* flag, unsigned v, hipStream_t s);
int launch_flag_probe(const unsigned* flag, unsigned want, unsigned* seen, hipStream_t s);  // gpk_init: are cross-stream flags usable here?

// ---------------------------------------------------------------- builders
struct SEParams {
  int d;
  double sf2, diag_add;   // diag_add = sn2 + jitter (Kxx only)
  double ell[32];        // length scales M (GPr.py:95); inputs are divided by them as GPr.py:100 does
};
// out[r][c] (ld) for r < nr_pad, c < nc_pad; R (nr x d), Cc (nc x d) device row-major.
// mode 0: sf2*exp(-0.5*r2) (+diag_add where global r == c if same_set); padded entries: identity if same_set else 0
// mode 1: plain squared distance r2 (scaled by inv_ell, pass ones for raw)
// The scalar-operand form of the builder (mode 0, d <= 8; k_build.hip): a pre-pass stores the centred, scaled points and their
// norms, the tile kernel takes the row point of an output row as SGPR operands from scalar loads (no LDS, no barrier) and its
// two column points ready-scaled from global memory.  The same bits as the LDS form.  Zero-initialised = the LDS form.
struct SEScalarForm {
  double* scratch = nullptr;           // se_scratch_doubles(...) doubles, the call's until the stream has passed the build
  int rows_ready = 0;                  // an earlier call on this stream with the same R, nr_pad, p and centre has left the scaled rows there
  const double* rows_norms = nullptr;  // (with rows_ready) the rows are a sub-range of a larger scaled set: scratch = its first row, the norms here
  const double* cols_scaled = nullptr; // column points scaled by an earlier call (the rows part of ITS scratch, same p and centre) ...
  int cols_pad = 0;                    // ... of that many points (norms behind the points)
};
inline size_t se_scratch_doubles(int nr_pad, int nc_pad, int d, bool cols_need_scaling) {
  return (size_t)nr_pad * (d + 1) + (cols_need_scaling ? (size_t)nc_pad * (d + 1) : 0);
}
// the pre-pass alone: S = the n_pad x d scaled points (zero rows beyond n), then their n_pad norms (d <= 8)
int launch_se_scale(const double* P, int n, int n_pad, const SEParams& p, const double* centre, double* S, hipStream_t s);
int launch_se_build(const double* R, int nr, int nr_pad, const double* Cc, int nc, int nc_pad, const SEParams& p,
                    int same_set, int mode, double* out, long ld, long lower_only, hipStream_t s, int col_g0 = 0,
                    const double* centre = nullptr, const SEScalarForm& form = SEScalarForm());
// centre (d doubles on the device): the point both sets are centred on in the dot-product form (d <= 8); nullptr = the
// call's first column point.  Builds of one model in several calls pass the same centre (the same bits in every tile).

// ---------------------------------------------------------------- aux kernels
int launch_sum_log_diag(const double* Dinv, int N, double* out, hipStream_t s);              // -sum log Dinv_ii, i < N
int launch_dot(const double* a, const double* b, int n, double* out, hipStream_t s);          // deterministic
int launch_row_stats(const double* X, long ld, int rows, int n, const double* w, double sf2, double* mean,
                     double* var, hipStream_t s);  // mean_j = X_j . w ; var_j = sf2 - X_j . X_j
// y = beta*y + alpha*A x  (one wave per row, coalesced);  gemv_sub: y -= A x
int launch_gemv(const double* A, long lda, int m, int k, const double* x, double* y, double alpha, double beta, hipStream_t s);
int launch_gemv_sub(const double* A, long lda, int m, int k, const double* x, double* y, hipStream_t s);
// y[0:k) -= A[m x k]^T x[0:m)  -- column sums, deterministic two-stage (ws >= 64*k doubles)
int launch_gemvT_sub(const double* A, long lda, int m, int k, const double* x, double* y, double* ws, hipStream_t s);
// diag-block solves for one panel of width w (multiple of 128) at L (leading dimension ld): forward x = Lpp^-1 b ; backward x = Lpp^-T b
int launch_diag_solve(const double* Lp, int w, long ld, const double* Dinv, double* b, int transpose, hipStream_t s);
int launch_transpose(const double* in, long ldin, int rows, int cols, double* out, long ldout, hipStream_t s);
int launch_fill(double* p, size_t n, double v, hipStream_t s);
int launch_stamp(unsigned long long* out, hipStream_t s);
int launch_spin(unsigned long long ticks, hipStream_t s);  // one wave spinning for `ticks` of the 100 MHz clock (stream-layout probe)
int launch_set_diag(double* p, long ld, int n, double v, hipStream_t s);
int launch_unpack_L(const double* Lp, int w, long ld, int h, int c0, double* out, long ldout, hipStream_t s);
int launch_gather2d(const double* K, long ldk, const long long* idx, int nidx, double* out, hipStream_t s);

// ---------------------------------------------------------------- preference-GP Laplace kernels (k_pref.hip)
int launch_pref_pair_stats(const long long* uvi, const double* y, const double* f, int npairs, double isq, double* z,
                           double* q, double* c, double* logphi, hipStream_t s);
int launch_pref_pair_t(const long long* uvi, const double* f, const double* c, const double* q, int npairs, double* t, hipStream_t s);
int launch_pref_scatter(const long long* uvi, const double* t, const double* scale, int do_sqrt, int npairs, double* out, hipStream_t s);
int launch_pref_gather(const long long* uvi, const double* x, const double* c, int npairs, double* r, hipStream_t s);
int launch_pref_gather_rows(const long long* uvi, const double* R, long ldr, const double* c, int npairs, int rows,
                             int rows_pad, int npad, double* U, long ldu, hipStream_t s);
int launch_pref_build_S(const double* K, long ldk, const long long* uvi, int npairs, double* S, long lds_, hipStream_t s);
int launch_pref_build_B(const double* S, long lds_, const double* c, int npairs, int npad, double* B, long ldb, hipStream_t s);
int launch_maxabsdiff(const double* a, const double* b, int n, double* out, hipStream_t s);
int launch_sum(const double* a, int n, double* out, hipStream_t s);
int launch_sub(const double* a, const double* b, int n, double* out, hipStream_t s);

// ---------------------------------------------------------------- classification Laplace kernels (k_class.hip)
int launch_class_stats(const double* y, const double* f, int n, int link, double* g, double* w, double* sw, double* logp,
                       hipStream_t s);
int launch_class_ew(const double* p, const double* q, const double* r, int n, int mode, double* out, hipStream_t s);
int launch_class_scale_rows(const double* R, long ldr, const double* sw, int n, int rows, int rows_pad, int npad, double* U,
                            long ldu, hipStream_t s);
int launch_class_prob(const double* mean, const double* var, int n, int link, double* prob, hipStream_t s);

// nlml gradient terms over A = Kinv - alpha alpha^T, NK = -Kinv (lower triangle valid); partial >= nt*nt*(d+2)
int launch_nlml_grad_terms(const double* X, int N, int d, const double* NK, long ldk, const double* alpha, const SEParams& p,
                           double* partial, double* out, hipStream_t s);
