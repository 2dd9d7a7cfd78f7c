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
};
// tile128 = true: 128x128 block tile (4x4 MFMA tiles per wave); false: 64x64 (2x2 per wave).
int launch_gemm_nt(const GemmNT& g, bool tile128, hipStream_t s);
// the 8-wave kernel, with the XCD-aware tile order for the largest updates
int launch_gemm_auto(const GemmNT& g, hipStream_t s);
// T1 (one full 128x128 tile) and T2 (one lower 128x128 tile that reads T1's result) of a tile-pipeline step in one launch;
// *ctr must have been `target - 4` when the launch starts (it counts T1's four blocks)
int launch_tt_fused(const GemmNT& t1, const GemmNT& t2, unsigned* ctr, unsigned target, int* info, hipStream_t s);
// flag-chained tile pipeline (potrf_tail, option tail_flags): the T step launched on the update stream ahead of its leaf,
// and the leaf kernel that waits for the previous T step's flag.  ctr2: two zeroed words, flags: zeroed words.
int launch_tt_flag(const GemmNT& t1, const GemmNT& t2, const unsigned* leaf_flag, unsigned* ctr2, unsigned* tt_flag, int* info, hipStream_t s);
int launch_leaf_flag(const double* A, long lda, double* L, long ldl, double* Dinv, int* info, int col0, const unsigned* wait_flag,
                     unsigned* set_flag, hipStream_t s);

// ---------------------------------------------------------------- leaf factor
// Factor the 128x128 diagonal tile A (lda) -> L tile (ldl, strict upper zeroed) and
// Dinv = inv(L) (dense 128x128, ld 128, strict upper zeroed).  *info gets
// (col0 + j + 1) of the first non-positive pivot (only if it is still 0).
int launch_potf2_trtri_128(const double* A, long lda, double* L, long ldl, double* Dinv, int* info,
                           int col0, int n_valid, hipStream_t s);

// The resident leaf server.  One 1024-thread workgroup (the leaf's own shape: a whole CU) is launched once per
// factorisation on a stream of its own and stays on its CU; the panel chain reaches it through a mailbox in HBM: a
// one-wave PROXY kernel in the chain stream posts the tile and spins until the server reports it done, so stream order
// and every dependency of the chain stay what they are with a direct launch.  What it buys: the leaf needs an EMPTY
// CU (132 KiB of LDS), which beside thousands of queued GEMM workgroups only a CU mask on the update stream could
// guarantee -- and a CU-masked queue places its waves unevenly over the SIMDs (scripts/mfma_pinned.py: 5/3 and 6/2
// waves per SIMD instead of 4/4), which costs the bulk update ~10 %.  With the server nothing is masked.
// Every wait is bounded (LEAF_SERVER_LIMIT_TICKS of the 100 MHz clock): a server that hears nothing exits, a proxy
// whose server does not answer gives up, both raise `err`, and the proxy poisons *info (LEAF_SERVER_INFO_POISON).
struct LeafMailbox {
  const double* A; long lda;
  double* L; long ldl;
  double* Dinv;
  int* info;
  int col0, op;              // op 0: factor a tile, 1: quit
  unsigned pad0[18];
  unsigned go;  unsigned pad1[31];    // sequence number of the posted task (own 128-byte line)
  unsigned done; unsigned pad2[31];   // sequence number of the last finished task
  unsigned err;  unsigned pad3[31];   // 1: the server timed out, 2: a proxy timed out
};
constexpr unsigned long long LEAF_SERVER_LIMIT_TICKS = 500000000ull;  // 5 s
constexpr int LEAF_SERVER_INFO_POISON = 0x40000000;
int launch_leaf_server(LeafMailbox* mb, unsigned first_seq, hipStream_t s);
int launch_leaf_proxy(LeafMailbox* mb, unsigned seq, int op, const double* A, long lda, double* L, long ldl,
                      double* Dinv, int* info, int col0, hipStream_t s);

// ---------------------------------------------------------------- builders
struct SEParams {
  int d;
  double sf2, diag_add;   // diag_add = sn2 + jitter (Kxx only)
  double ell[32];        // length scales M (GPr.py:95); inputs are divided by them as GPr.py:100 does
};
// out[r][c] (ld) for r < nr_pad, c < nc_pad; R (nr x d), Cc (nc x d) device row-major.
// mode 0: sf2*exp(-0.5*r2) (+diag_add where global r == c if same_set); padded entries: identity if same_set else 0
// mode 1: plain squared distance r2 (scaled by inv_ell, pass ones for raw)
int launch_se_build(const double* R, int nr, int nr_pad, const double* Cc, int nc, int nc_pad, const SEParams& p,
                    int same_set, int mode, double* out, long ld, long lower_only, hipStream_t s, int col_g0 = 0);

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
