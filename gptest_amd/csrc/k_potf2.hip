// Leaf of the blocked Cholesky: factor one 128x128 diagonal tile and invert the triangular
// factor (the inverse turns every panel solve into an MFMA GEMM).  Replaces the unblocked
// interior of LAPACK potrf behind np.linalg.cholesky (GPr.py:62, GPpref.py:128).
//
// One workgroup on one CU, two forms of the SAME blocked algorithm (bit-identical results):
//   * the slot-sized leaf (512 threads, 76.6 KiB block-triangular LDS image): fits the slot one finished trailing-update
//     workgroup frees, announces itself in the guest table so that the GEMM workgroups on its CU pause (gpk_internal.h);
//   * the full-tile leaf (1024 threads, 132 KiB): alone on its CU, 2 us faster, for when an empty CU is easy to find.
//  1. Blocked elimination, 8 steps of 16 columns on the tile in LDS: the 16x16 diagonal block is factored by one wave inside
//     ONE MFMA accumulator, the rows below solve against it by substitution (one thread per row), the trailing 16x16 blocks
//     take the rank-16 update on the matrix pipe while wave 0 already factors the next diagonal block.
//  2. In-place inverse of L: the 16x16 diagonal blocks by per-wave forward substitution, then recursive doubling
//     X21 = -X22*(L21*X11) for block sizes 16/32/64 with v_mfma_f64_16x16x4_f64 reading its fragments
//     straight from the tile.
#include "gpk_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ double readlane_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

#define PLD 129
#define LEAF_STAMP(k) do { if ((dbg_skip & 8) && tid == 0) info[k] = (int)__builtin_amdgcn_s_memrealtime(); } while (0)


// Factor of the 16x16 diagonal block at column c0 of the LDS tile, executed by ONE wave, rank-4 updates on the
// matrix pipe.  The 16x16 block lives in ONE accumulator (C/D layout:
// lane (r = lane & 15, k = lane >> 4), register q holds element (row k + 4q, col r)) and stays fully symmetric, so
// register q of lane (r, k) IS the panel element (row r, column 4q + k) in MFMA operand layout: no data movement
// between "take the next four columns" and "apply them".  Per four columns: gather the panel row-wise (4
// ds_bpermute pairs), factor the four columns with readlane broadcasts (6 updates instead of 54), one
// v_mfma_f64_16x16x4_f64 for the whole trailing block.  ~100 instructions per four columns against ~240 for the
// lane-per-row version with 15 readlane broadcasts per column (measured 18.5 vs 19.4 us per leaf: the chain of
// dependent fp64 operations per pivot, not the instruction count, is what is left -- round 4 took that chain from 17
// dependent operations to 12, below: 18.4 -> 13.8 us per leaf).
// Sb: element (0,0) of the block, LD: its row stride in doubles (the full-tile image of the 1024-thread leaf and the
// block-triangular image of the slot-sized leaf share this code: same operations, same order, same bits).
template <int LD>
static __device__ __forceinline__ void leaf_diag_block_mfma(double* Sb, int c0, int lane, double* rinvb, int* failflag) {
  const int r = lane & 15, k = lane >> 4;
  d4 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = k + 4 * q;  // only the lower triangle of the tile is read (LAPACK convention), mirrored
    acc[q] = (row >= r) ? Sb[row * LD + r] : Sb[r * LD + row];
  }
  int bad = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int jb = 4 * q;
    const double praw = acc[q];
    double p[4], l[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) p[c] = __shfl(praw, r + 16 * c, 64);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double piv = readlane_d(p[c], jb + c);
      const double p2 = p[c] + p[c];  // (beside the root's chain, not on it)
      const bool ok = (piv > 1.0e-300) && (piv < 1.0e300);
      if (!ok && !bad) bad = jb + c + 1;
      // sqrt(piv) and 0.5 / sqrt(piv) by the coupled (Goldschmidt) iteration from v_rsq_f64: TWO dependent operations per step
      // where the Newton form of rounds 1-3 had three, and the root itself comes with it instead of behind it; the column below
      // the pivot and its broadcast to the next columns do not wait for the diagonal element either (`lo`: rows above the
      // pivot hold values nobody reads -- every later use of p[c2] and of l[] masks them).  12 dependent operations per pivot
      // against 17: 144 -> 108 ns, on every leaf of every chain step (full-tile leaf 49.4 -> 44.9 us, slot-sized 51.5 -> 46.8).
      // Accuracy (scripts/leaf_accuracy.py): the root within 1 ulp, 1 / root within 2 ulp of the correctly rounded values over 16
      // decades; residual of a dense tile 3e-16.
      const double y0 = __builtin_amdgcn_rsq(piv);
      double g = piv * y0, h = 0.5 * y0;
      double e = __builtin_fma(-g, h, 0.5);
      g = __builtin_fma(g, e, g);
      h = __builtin_fma(h, e, h);
      e = __builtin_fma(-g, h, 0.5);
      g = __builtin_fma(g, e, g);
      h = __builtin_fma(h, e, h);
      const double lo = p2 * h;
      l[c] = (r == jb + c) ? g : ((r > jb + c) ? lo : 0.0);
      if (lane == 0) rinvb[jb + c] = h + h;
#pragma unroll
      for (int c2 = c + 1; c2 < 4; ++c2) {
        const double lkj = readlane_d(lo, jb + c2);
        p[c2] = __builtin_fma(-lo, lkj, p[c2]);
      }
    }
    if (lane < 16) {
      double* wrow = Sb + r * LD + jb;
#pragma unroll
      for (int c = 0; c < 4; ++c) wrow[c] = l[c];
    }
    if (q < 3) {
      const double pk = (k == 0) ? l[0] : (k == 1) ? l[1] : (k == 2) ? l[2] : l[3];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pk, pk, acc, 0, 0, 0);
    }
  }
  if (bad && lane == 0) *failflag = c0 + bad;
}
// rank-16 update of trailing tile t (triangular numbering, tile 0 = the next diagonal block) on the matrix pipe
static __device__ __forceinline__ void leaf_update_tile(double* S, int c0, int t, int fr, int fg) {
  int I = 0;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int J = t - I * (I + 1) / 2;
  const int RI = c0 + 16 + 16 * I, RJ = c0 + 16 + 16 * J;
  d4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = S[(RI + fg + 4 * r) * PLD + RJ + fr];
  const double* Ap = S + (RI + fr) * PLD + c0 + fg;
  const double* Bp = S + (RJ + fr) * PLD + c0 + fg;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[4 * ks], Bp[4 * ks], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) S[(RI + fg + 4 * r) * PLD + RJ + fr] = acc[r];
}

// The full-tile leaf's work on one tile.  S (128 * PLD doubles, 16-byte aligned), rinvb (16) and failflag live in the
// calling kernel's LDS.  dbg_skip (timing only): 1 no elimination, 2 no inverse, 8 phase stamps in info[3..7], 16 no panel
// solve, 32 no trailing updates, 64 no diagonal blocks.
static __device__ __forceinline__ void leaf_body(const double* __restrict__ A, long lda, double* __restrict__ L, long ldl,
                                                 double* __restrict__ Dinv, int* info, int col0, int dbg_skip,
                                                 double* S, double* rinvb, int& failflag, const int tid) {
  LEAF_STAMP(3);
  {  // (all sixteen loads in flight before the first LDS store, as in leaf_body_tri)
    double v[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int e = tid + 1024 * p;
      v[p] = A[(long)(e >> 7) * lda + (e & 127)];
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int e = tid + 1024 * p;
      S[(e >> 7) * PLD + (e & 127)] = v[p];
    }
  }
  __syncthreads();
  LEAF_STAMP(4);
  {
    // ---- blocked elimination: 8 steps of 16 columns, the tile stays in LDS.
    //  (1) wave 0 factors the 16x16 diagonal block in one MFMA accumulator (leaf_diag_block_mfma: no barrier and no
    //      LDS round trip inside the 16 sequential columns), rsqrt + Newton on the pivot;
    //  (2) the rows below solve against it by substitution, one thread per row, L11 read as LDS broadcasts;
    //  (3) the trailing 16x16 tiles take the rank-16 update on the matrix pipe, one tile per wave and round, while
    //      wave 0, after updating the next diagonal tile, already factors it (look-ahead).
    // Two barriers per 16 columns instead of one per column.
    const int lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    if (tid == 0) failflag = 0;
    __syncthreads();
    int fail = 0;
    const bool run = !(dbg_skip & 1);
    if (run && wv == 0 && !(dbg_skip & 64)) leaf_diag_block_mfma<PLD>(S, 0, lane, rinvb, &failflag);
    __syncthreads();
    fail = failflag;
    for (int b = 0; b < 8 && run && !fail; ++b) {
      const int c0 = 16 * b;
      const int nbelow = 112 - c0;
      if (tid < nbelow && !(dbg_skip & 16)) {
        double* xr = S + (c0 + 16 + tid) * PLD + c0;
        double x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = xr[k];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          x[m] *= rinvb[m];
#pragma unroll
          for (int k = m + 1; k < 16; ++k) x[k] = __builtin_fma(-x[m], S[(c0 + k) * PLD + c0 + m], x[k]);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) xr[k] = x[k];
      }
      __syncthreads();
      if (b == 7) break;
      // look-ahead: wave 0 updates the next diagonal tile and factors it at once (2.1 us) while the other
      // fifteen waves apply the rest of this step's update (<= 27 tiles)
      const int nbk = 7 - b, ntile = nbk * (nbk + 1) / 2;
      if (wv == 0) {
        if (!(dbg_skip & 32)) leaf_update_tile(S, c0, 0, fr, fg);
        if (!(dbg_skip & 64)) leaf_diag_block_mfma<PLD>(S + (c0 + 16) * PLD + c0 + 16, c0 + 16, lane, rinvb, &failflag);
      } else {
        for (int t = wv; t < ntile && !(dbg_skip & 32); t += 15) leaf_update_tile(S, c0, t, fr, fg);
      }
      __syncthreads();
      fail = failflag;
    }
    if (fail) {
      if (tid == 0) atomicCAS(info, 0, col0 + fail);
      for (int e = tid; e < 128 * 128; e += 1024) {
        const int ii = e >> 7, jj = e & 127;
        const double v = (ii == jj) ? 1.0 : 0.0;
        L[(long)ii * ldl + jj] = v;
        Dinv[ii * 128 + jj] = v;
      }
      return;
    }
    for (int e = tid; e < 128 * 128; e += 1024) {  // blocks above the diagonal still hold updated symmetric data
      const int ii = e >> 7, jj = e & 127;
      if (jj > ii) S[ii * PLD + jj] = 0.0;
    }
    __syncthreads();
  }
  for (int e = tid; e < 128 * 128; e += 1024) {
    const int ii = e >> 7, jj = e & 127;
    L[(long)ii * ldl + jj] = S[ii * PLD + jj];
  }
  __syncthreads();
  if (tid < 128) S[tid * PLD + tid] = 1.0 / S[tid * PLD + tid];
  __syncthreads();
  LEAF_STAMP(5);

  // ---- inverses of the eight 16x16 diagonal blocks: one wave per block, lane = column of X = L11^-1, the column
  // is built in registers by forward substitution (L11 rows arrive as LDS broadcasts; the diagonal already holds
  // reciprocals) and then written over the block.  No barrier inside; replaces four levels of recursive doubling
  // (block sizes 1..8, sixteen barriers) at the same cost.
  if (!(dbg_skip & 2)) {
    const int w16 = tid >> 6, c = tid & 63;
    if (w16 < 8 && c < 16) {
      const int c0 = 16 * w16;
      double x[16];
#pragma unroll
      for (int ii = 0; ii < 16; ++ii) {
        const double* Li = S + (c0 + ii) * PLD + c0;
        const double rii = Li[ii];
        double sum = 0.0;
#pragma unroll
        for (int m = 0; m < ii; ++m) sum = __builtin_fma(Li[m], x[m], sum);  // x[m] = 0 for m < c
        x[ii] = (ii == c) ? rii : ((ii > c) ? -sum * rii : 0.0);
        asm volatile("" : "+v"(x[ii]) : : "memory");  // finish row ii before the reads of row ii+1 are issued: without
                                                       // it all 120 broadcast reads are hoisted and the kernel spills
      }
#pragma unroll
      for (int ii = 0; ii < 16; ++ii)
        if (ii >= c) S[(c0 + ii) * PLD + c0 + c] = x[ii];
    }
    __syncthreads();
  }
  // ---- block sizes 16, 32, 64: MFMA.  b/4 output tiles of 16x16 per phase, one per wave.
  const int lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int nlev_mfma = (dbg_skip & 2) ? 0 : 3;
  for (int lv = 0; lv < nlev_mfma; ++lv) {
    const int b = 16 << lv;
    const int tpp = (b / 16) * (b / 16);  // tiles per pair
    const int ntile = (64 / b) * tpp;     // 4, 8, 16
    const bool act = wv < ntile;
    int r0 = 0, trow = 0, tcol = 0;
    if (act) {
      const int pair = wv / tpp, tt = wv % tpp;
      r0 = 2 * b * pair;
      trow = (tt / (b / 16)) * 16;
      tcol = (tt % (b / 16)) * 16;
    }
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    if (act) {
      // T = L21 * X11 : A = L21[trow + fr][k + fg], B = X11[k + fg][tcol + fr]  (zeros above the diagonal)
      const double* Ap = S + (r0 + b + trow + fr) * PLD + r0 + fg;
      const double* Bp = S + (r0 + fg) * PLD + r0 + tcol + fr;
      for (int k = 0; k < b; k += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ap[k], Bp[k * PLD], acc, 0, 0, 0);
    }
    __syncthreads();
    if (act) {
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(r0 + b + trow + fg + 4 * r) * PLD + r0 + tcol + fr] = acc[r];
    }
    __syncthreads();
    acc = (d4){0.0, 0.0, 0.0, 0.0};
    if (act) {
      // X21 = -X22 * T : A = X22[trow + fr][k + fg], B = T[k + fg][tcol + fr]
      const double* Ap = S + (r0 + b + trow + fr) * PLD + r0 + b + fg;
      const double* Bp = S + (r0 + b + fg) * PLD + r0 + tcol + fr;
      for (int k = 0; k < b; k += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ap[k], Bp[k * PLD], acc, 0, 0, 0);
    }
    __syncthreads();
    if (act) {
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(r0 + b + trow + fg + 4 * r) * PLD + r0 + tcol + fr] = -acc[r];
    }
    __syncthreads();
  }
  LEAF_STAMP(6);
  for (int e = tid; e < 128 * 128; e += 1024) {
    const int ii = e >> 7, jj = e & 127;
    Dinv[e] = (jj <= ii) ? S[ii * PLD + jj] : 0.0;
  }
  LEAF_STAMP(7);
}


// ---------------------------------------------------------------- the slot-sized leaf (default)
// The same blocked algorithm -- operation for operation, so the factor and its inverse are BIT-IDENTICAL to leaf_body --
// in a workgroup that fits the slot ONE finished trailing-update workgroup frees on a CU: 8 waves of <= 128 VGPRs and 76.6 KiB
// of LDS (the update kernel holds 64 KiB per workgroup, two per CU, of 160 KiB).  The 1024-thread / 132 KiB form needs an
// EMPTY CU, which beside thousands of queued update workgroups only a CU mask on the update stream could provide, and a
// CU-masked queue costs the update kernel ~10 %.  With this form nothing is masked: the leaf is
// launched on the high-priority chain stream and takes the next slot that frees up anywhere on the chip.
// LDS image: only the 36 lower 16x16 blocks, block (I, J) at tri_base(I, J) (block-column-major: blocks of one block column
// are consecutive, so that consecutive block rows sit an odd multiple of 16 doubles apart), rows 17 doubles apart.
#define TLD 17
#define TBS (16 * TLD)
#define TRI_DOUBLES (36 * TBS)
static __device__ __forceinline__ int tri_base(int I, int J) { return (8 * J - J * (J - 1) / 2 + (I - J)) * TBS; }

// rank-16 update of trailing block t of step b (triangular numbering, block 0 = the next diagonal block)
static __device__ __forceinline__ void leaf_update_tile_tri(double* S, int b, int t, int fr, int fg) {
  int I = 0;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int J = t - I * (I + 1) / 2;
  double* C = S + tri_base(b + 1 + I, b + 1 + J);
  d4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = C[(fg + 4 * r) * TLD + fr];
  const double* Ap = S + tri_base(b + 1 + I, b) + fr * TLD + fg;
  const double* Bp = S + tri_base(b + 1 + J, b) + fr * TLD + fg;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[4 * ks], Bp[4 * ks], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) C[(fg + 4 * r) * TLD + fr] = acc[r];
}

// S: TRI_DOUBLES doubles of LDS; 512 threads.  dbg_skip: 1 no elimination, 2 no inverse, 8 phase stamps in info[3..7].
static __device__ __forceinline__ void leaf_body_tri(const double* __restrict__ A, long lda, double* __restrict__ L, long ldl,
                                                     double* __restrict__ Dinv, int* info, int col0, int dbg_skip, double* S,
                                                     double* rinvb, int& failflag, const int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  LEAF_STAMP(3);
  // ---- load the lower blocks: block row I is 16 rows of 16 (I + 1) contiguous doubles
  // (16-byte loads: A is 16-byte aligned with an even leading dimension, as every operand of the GEMM kernels)
  // ALL twelve loads of a thread in flight before the first LDS store: written as one loop the compiler waited for every load
  // before its store -- twelve dependent round trips, 6 us alone and 20-30 beside an update (the leaf's 47 -> 72 us in situ)
  {
    d2 v[12];
#pragma unroll
    for (int I = 0, k = 0; I < 8; ++I) {
      const int rl = 8 * (I + 1);  // pairs per row
#pragma unroll
      for (int p = 0; p < (16 * rl + 511) / 512; ++p, ++k) {
        const int e = tid + 512 * p;
        const int r = e / rl, c = 2 * (e - r * rl);
        v[k] = (e < 16 * rl) ? *reinterpret_cast<const d2*>(A + (long)(16 * I + r) * lda + c) : d2{0.0, 0.0};
      }
    }
#pragma unroll
    for (int I = 0, k = 0; I < 8; ++I) {
      const int rl = 8 * (I + 1);
#pragma unroll
      for (int p = 0; p < (16 * rl + 511) / 512; ++p, ++k) {
        const int e = tid + 512 * p;
        const int r = e / rl, c = 2 * (e - r * rl);
        if (e < 16 * rl) {
          double* dst = S + tri_base(I, c >> 4) + r * TLD + (c & 15);
          dst[0] = v[k].x;
          dst[1] = v[k].y;
        }
      }
    }
  }
  if (tid == 0) failflag = 0;
  __syncthreads();
  LEAF_STAMP(4);
  // ---- blocked elimination: as leaf_body<1> (8 steps of 16 columns; wave 0 updates and factors the next diagonal block
  // while waves 1..7 apply the rest of the step's rank-16 update)
  int fail = 0;
  const bool run = !(dbg_skip & 1);
  if (run && wv == 0) leaf_diag_block_mfma<TLD>(S + tri_base(0, 0), 0, lane, rinvb, &failflag);
  __syncthreads();
  fail = failflag;
  for (int b = 0; b < 8 && run && !fail; ++b) {
    const int nbelow = 112 - 16 * b;
    if (tid < nbelow) {
      double* xr = S + tri_base(b + 1 + (tid >> 4), b) + (tid & 15) * TLD;
      const double* Sd = S + tri_base(b, b);
      double x[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = xr[k];
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        x[m] *= rinvb[m];
#pragma unroll
        for (int k = m + 1; k < 16; ++k) x[k] = __builtin_fma(-x[m], Sd[k * TLD + m], x[k]);
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) xr[k] = x[k];
    }
    __syncthreads();
    if (b == 7) break;
    const int nbk = 7 - b, ntile = nbk * (nbk + 1) / 2;
    if (wv == 0) {
      leaf_update_tile_tri(S, b, 0, fr, fg);
      leaf_diag_block_mfma<TLD>(S + tri_base(b + 1, b + 1), 16 * b + 16, lane, rinvb, &failflag);
    } else {
      for (int t = wv; t < ntile; t += 7) leaf_update_tile_tri(S, b, t, fr, fg);
    }
    __syncthreads();
    fail = failflag;
  }
  if (fail) {
    if (tid == 0) atomicCAS(info, 0, col0 + fail);
    // keep downstream kernels on finite data: emit identity factors for this tile
    for (int e = tid; e < 128 * 128; e += 512) {
      const int ii = e >> 7, jj = e & 127;
      const double v = (ii == jj) ? 1.0 : 0.0;
      L[(long)ii * ldl + jj] = v;
      Dinv[ii * 128 + jj] = v;
    }
    return;
  }
  for (int e = tid; e < 128 * 64; e += 512) {
    const int ii = e >> 6, jj = (e & 63) * 2;
    d2 v = {0.0, 0.0};
    if ((jj >> 4) <= (ii >> 4)) {
      const double* src = S + tri_base(ii >> 4, jj >> 4) + (ii & 15) * TLD + (jj & 15);
      v = (d2){src[0], src[1]};
    }
    *reinterpret_cast<d2*>(L + (long)ii * ldl + jj) = v;
  }
  __syncthreads();
  if (tid < 128) {
    double* dg = S + tri_base(tid >> 4, tid >> 4) + (tid & 15) * (TLD + 1);
    *dg = 1.0 / *dg;
  }
  __syncthreads();
  LEAF_STAMP(5);
  if (!(dbg_skip & 2)) {
    // ---- inverses of the eight diagonal blocks: one wave per block, lane = column (as leaf_body)
    if (lane < 16) {
      double* Sd = S + tri_base(wv, wv);
      const int c = lane;
      double x[16];
#pragma unroll
      for (int ii = 0; ii < 16; ++ii) {
        const double* Li = Sd + ii * TLD;
        const double rii = Li[ii];
        double sum = 0.0;
#pragma unroll
        for (int m = 0; m < ii; ++m) sum = __builtin_fma(Li[m], x[m], sum);
        x[ii] = (ii == c) ? rii : ((ii > c) ? -sum * rii : 0.0);
        asm volatile("" : "+v"(x[ii]) : : "memory");
      }
#pragma unroll
      for (int ii = 0; ii < 16; ++ii)
        if (ii >= c) Sd[ii * TLD + c] = x[ii];
    }
    __syncthreads();
    // ---- recursive doubling X21 = -X22 (L21 X11) for block sizes 16, 32, 64: up to 16 output blocks per level, two per
    // wave.  The zero blocks above the diagonal of X11 / X22 (which the full-tile image multiplies by) are skipped: the
    // same sums.
#pragma unroll
    for (int lv = 0; lv < 3; ++lv) {
      const int nb = 1 << lv;          // level block size in 16-blocks
      const int tpp = nb * nb;         // output blocks per pair
      const int ntile = (4 >> lv) * tpp;  // 4, 8, 16
      d4 acc[2];
      int R[2], Cb[2], K0[2], tr[2], tc[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int tile = wv + 8 * u;
        acc[u] = (d4){0.0, 0.0, 0.0, 0.0};
        if (tile < ntile) {
          const int pair = tile / tpp, tt = tile % tpp;
          K0[u] = 2 * nb * pair;       // first 16-block of X11
          tr[u] = tt / nb;
          tc[u] = tt % nb;
          R[u] = K0[u] + nb + tr[u];
          Cb[u] = K0[u] + tc[u];
          // T = L21 X11: A = L21[fr][k + fg] from block (R, K0 + kb), B = X11[k + fg][fr] from block (K0 + kb, Cb), kb >= tc
          for (int kb = tc[u]; kb < nb; ++kb) {
            const double* Ap = S + tri_base(R[u], K0[u] + kb) + fr * TLD + fg;
            const double* Bp = S + tri_base(K0[u] + kb, Cb[u]) + fg * TLD + fr;
#pragma unroll
            for (int k = 0; k < 16; k += 4) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ap[k], Bp[k * TLD], acc[u], 0, 0, 0);
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (wv + 8 * u < ntile) {
          double* T = S + tri_base(R[u], Cb[u]);
#pragma unroll
          for (int r = 0; r < 4; ++r) T[(fg + 4 * r) * TLD + fr] = acc[u][r];
        }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        acc[u] = (d4){0.0, 0.0, 0.0, 0.0};
        if (wv + 8 * u < ntile) {
          // X21 = -X22 T: A = X22[fr][k + fg] from block (R, K0 + nb + kb), kb <= tr; B = T[k + fg][fr] from block (K0 + nb + kb, Cb)
          for (int kb = 0; kb <= tr[u]; ++kb) {
            const double* Ap = S + tri_base(R[u], K0[u] + nb + kb) + fr * TLD + fg;
            const double* Bp = S + tri_base(K0[u] + nb + kb, Cb[u]) + fg * TLD + fr;
#pragma unroll
            for (int k = 0; k < 16; k += 4) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ap[k], Bp[k * TLD], acc[u], 0, 0, 0);
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (wv + 8 * u < ntile) {
          double* X = S + tri_base(R[u], Cb[u]);
#pragma unroll
          for (int r = 0; r < 4; ++r) X[(fg + 4 * r) * TLD + fr] = -acc[u][r];
        }
      __syncthreads();
    }
  }
  LEAF_STAMP(6);
  for (int e = tid; e < 128 * 64; e += 512) {
    const int ii = e >> 6, jj = (e & 63) * 2;
    d2 v = {0.0, 0.0};
    if ((jj >> 4) <= (ii >> 4)) {  // (the diagonal blocks hold zeros above the diagonal: leaf_diag_block_mfma wrote them, the inverse kept them)
      const double* src = S + tri_base(ii >> 4, jj >> 4) + (ii & 15) * TLD + (jj & 15);
      v = (d2){src[0], (jj + 1 <= ii) ? src[1] : 0.0};
    }
    *reinterpret_cast<d2*>(Dinv + (long)ii * 128 + jj) = v;
  }
  LEAF_STAMP(7);
}

__global__ __launch_bounds__(512, 4) void potf2_trtri_128_slot_kernel(const double* __restrict__ A, long lda, double* __restrict__ L,
                                                                      long ldl, double* __restrict__ Dinv, int* info, int col0,
                                                                      int dbg_skip, unsigned* guest) {
  __shared__ __attribute__((aligned(16))) double S[TRI_DOUBLES];
  __shared__ double rinvb[16];
  __shared__ int failflag;
  // announce the guest on this CU (gpk_internal.h: the GEMM workgroups that share the CU pause while the counter is up)
  unsigned* gslot = nullptr;
  if (guest && threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    gslot = guest + (((xcc & 7u) << 8) | ((hw >> 8) & 0xffu));
    __hip_atomic_fetch_add(gslot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  leaf_body_tri(A, lda, L, ldl, Dinv, info, col0, dbg_skip, S, rinvb, failflag, (int)threadIdx.x);
  __syncthreads();  // every path of the body ends here: the counter always comes down again
  if (gslot) __hip_atomic_fetch_sub(gslot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(1024) void potf2_trtri_128_kernel(const double* __restrict__ A, long lda,
                                                               double* __restrict__ L, long ldl,
                                                               double* __restrict__ Dinv, int* info, int col0,
                                                               int dbg_skip) {
  __shared__ __attribute__((aligned(16))) double S[128 * PLD];
  __shared__ double rinvb[16];
  __shared__ int failflag;
  leaf_body(A, lda, L, ldl, Dinv, info, col0, dbg_skip, S, rinvb, failflag, (int)threadIdx.x);
}

// ---------------------------------------------------------------- one launch per chain step (tile pipeline, potrf_tail)
// Step p of the tile pipeline used to be three dependent launches on the chain stream: the leaf of tile (p,p), then
// T1: L(p+1,p) = A(p+1,p) Dinv(p)^T and T2: A(p+1,p+1) -= L(p+1,p) L(p+1,p)^T as one-tile GEMMs (7 us each plus a launch gap
// each).  Here the full-tile leaf goes on, on its own CU, with both products: Dinv is already in LDS (B operand of T1), the
// rows of A(p+1,p) come straight from L2 (32 values per lane, one load latency), L(p+1,p) is written to the packed panel and
// into LDS (row stride CLD: conflict-free fragment reads) for T2.  Every output element is ONE MFMA chain over k in ascending
// order, started from zero (T1) or from the stored value (T2), exactly as in gemm_tile128_kernel -- the k-blocks of T1 beyond
// the diagonal of Dinv are exact zeros there and skipped here -- so the factor is bit for bit that of the three launches
// (tests).  The two tiles of row p+1 carry the look-ahead update of step p-1, which runs on the UPDATE stream while this
// kernel factors tile (p,p): instead of an event in front of the launch the kernel polls a flag that the update stream sets
// behind that update (launch_flag_set), after the leaf and before it touches row p+1.  Bounded: after GPK_CHAIN_WAIT_TICKS of
// the 100 MHz clock the kernel gives up and poisons *info (GPK_INFO_TIMEOUT -> an error on the host, never a hang).
#define CLD 146
constexpr unsigned long long GPK_CHAIN_WAIT_TICKS = 400000ull;  // 4 ms

template <int J0, int J1, int J2, int J3>
static __device__ __forceinline__ void chain_t1(const double (&a)[32], const double* S, int fr, int fg, d4 (&acc)[4]) {
  constexpr int Js[4] = {J0, J1, J2, J3};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    acc[u] = (d4){0.0, 0.0, 0.0, 0.0};
    const double* Bp = S + (16 * Js[u] + fr) * PLD + fg;  // rows of Dinv = columns of the product; Dinv[c][k] = 0 for k > c
#pragma unroll
    for (int q = 0; q < 4 * (Js[u] + 1); ++q) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], Bp[4 * q], acc[u], 0, 0, 0);
  }
}

__global__ __launch_bounds__(1024) void potf2_chain_kernel(double* __restrict__ A, long lda, double* __restrict__ L, long ldl,
                                                           double* __restrict__ Dinv, int* info, int col0, const unsigned* flag,
                                                           unsigned want, unsigned* done_leaf, unsigned* done_t1, int do_t2) {
  __shared__ __attribute__((aligned(16))) double S[128 * CLD];  // the leaf's 128 x PLD image first, the image of L(p+1,p) later
  __shared__ double rinvb[16];
  __shared__ int failflag;
  const int tid = (int)threadIdx.x;
  leaf_body(A, lda, L, ldl, Dinv, info, col0, 0, S, rinvb, failflag, tid);
  __syncthreads();
  if (failflag) {  // identity factors went out for this tile (leaf_body): the two products go on with them
    for (int e = tid; e < 128 * 128; e += 1024) S[(e >> 7) * PLD + (e & 127)] = ((e >> 7) == (e & 127)) ? 1.0 : 0.0;
    __syncthreads();
  }
  const int lane = tid & 63, wv = tid >> 6, fr = lane & 15, fg = lane >> 4;
  if (done_leaf) {
    // L(p,p) and Dinv(p) are out: the full-height solve of this step, already resident on the update stream and polling this
    // flag (GemmNT::wait_flag), can go on while this kernel does the two tile products
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(done_leaf, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (flag) {
    if (tid == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > GPK_CHAIN_WAIT_TICKS) {
          atomicCAS(info, 0, 1 << 30);  // GPK_INFO_TIMEOUT (gpk_core.h)
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // row p+1 was written by another kernel while this one was running
  }
  // ---- T1: L(p+1,p) = A(p+1,p) Dinv^T.  Wave (I, half): block row I, block columns {0,3,4,7} or {1,2,5,6} (18 k-blocks each).
  const int I = wv & 7, half = wv >> 3;
  const double* A1 = A + 128 * lda;
  double* C2 = A + 128 * lda + 128;
  double a[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) a[q] = A1[(long)(16 * I + fr) * lda + 4 * q + fg];
  d4 acc[4];
  if (half == 0) chain_t1<0, 3, 4, 7>(a, S, fr, fg, acc);
  else chain_t1<1, 2, 5, 6>(a, S, fr, fg, acc);
  // the accumulators of T2 start as the stored tile: requested now (the operand registers of T1 are free), needed two barriers on
  d4 c2[3];
  int ti2[3], tj2[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int t = wv + 16 * u;  // lower 16x16 blocks in triangular numbering, 36 of them over 16 waves
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    ti2[u] = bi;
    tj2[u] = t - bi * (bi + 1) / 2;
    if (t < 36 && do_t2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) c2[u][r] = C2[(long)(16 * ti2[u] + fg + 4 * r) * lda + 16 * tj2[u] + fr];
    }
  }
  __syncthreads();  // every wave has its fragments of Dinv: the image can go
  double* L1 = L + 128 * ldl;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int J = half == 0 ? (u == 0 ? 0 : u == 1 ? 3 : u == 2 ? 4 : 7) : (u == 0 ? 1 : u == 1 ? 2 : u == 2 ? 5 : 6);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * I + fg + 4 * r, col = 16 * J + fr;
      L1[(long)row * ldl + col] = acc[u][r];
      S[row * CLD + col] = acc[u][r];
    }
  }
  if (done_t1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // L(p+1,p) is out: the look-ahead update of this step may read it
  __syncthreads();
  if (done_t1 && tid == 0) __hip_atomic_store(done_t1, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  // ---- T2: A(p+1,p+1) -= L(p+1,p) L(p+1,p)^T, lower blocks (not at the end of a batch of the batched pipeline: tile (p+1,p+1) then
  //      belongs to the bulk stream's look-ahead update of the next batch)
  if (!do_t2) return;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    if (wv + 16 * u < 36) {
      const double* Ap = S + (16 * ti2[u] + fr) * CLD + fg;
      const double* Bp = S + (16 * tj2[u] + fr) * CLD + fg;
      d4 c = c2[u];
#pragma unroll
      for (int q = 0; q < 32; ++q) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[4 * q], Bp[4 * q], c, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) C2[(long)(16 * ti2[u] + fg + 4 * r) * lda + 16 * tj2[u] + fr] = c[r];
    }
  }
}

__global__ void flag_set_kernel(unsigned* flag, unsigned v) {
  __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
int launch_flag_set(unsigned* flag, unsigned v, hipStream_t s) {
  hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(1), 0, s, flag, v);
  GPK_HIP(hipGetLastError());
  return 0;
}
// probe (gpk_init): can a kernel on one stream see a flag that a LATER launch on another stream sets while it runs?  Under a tool
// that serialises dispatches (rocprofv3 --pmc) it cannot, and the fused chain step must not be used there.
__global__ void flag_probe_kernel(const unsigned* flag, unsigned want, unsigned* seen) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned ok = 0;
  do {
    if ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) { ok = 1; break; }
    __builtin_amdgcn_s_sleep(8);
  } while (__builtin_amdgcn_s_memrealtime() - t0 < 50000ull);  // 0.5 ms
  *seen = ok;
}
int launch_flag_probe(const unsigned* flag, unsigned want, unsigned* seen, hipStream_t s) {
  hipLaunchKernelGGL(flag_probe_kernel, dim3(1), dim3(1), 0, s, flag, want, seen);
  GPK_HIP(hipGetLastError());
  return 0;
}

int launch_potf2_chain(double* A, long lda, double* L, long ldl, double* Dinv, int* info, int col0, const unsigned* flag, unsigned want,
                       hipStream_t s, unsigned* done_leaf, unsigned* done_t1, int do_t2) {
  hipLaunchKernelGGL(potf2_chain_kernel, dim3(1), dim3(1024), 0, s, A, lda, L, ldl, Dinv, info, col0, flag, want, done_leaf, done_t1, do_t2);
  GPK_HIP(hipGetLastError());
  return 0;
}

// fat: the full-tile form (the caller expects an empty CU to be easy to find), else the slot-sized one
int launch_potf2_trtri_128(const double* A, long lda, double* L, long ldl, double* Dinv, int* info, int col0,
                           int dbg_skip, hipStream_t s, int fat) {
  if (!fat)
    hipLaunchKernelGGL(potf2_trtri_128_slot_kernel, dim3(1), dim3(512), 0, s, A, lda, L, ldl, Dinv, info, col0, dbg_skip, g_cu_guest);
  else
    hipLaunchKernelGGL(potf2_trtri_128_kernel, dim3(1), dim3(1024), 0, s, A, lda, L, ldl, Dinv, info, col0, dbg_skip);
  GPK_HIP(hipGetLastError());
  return 0;
}
