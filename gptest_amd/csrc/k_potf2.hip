// Leaf of the blocked Cholesky: factor one 128x128 diagonal tile and invert the triangular
// factor (the inverse turns every panel solve into an MFMA GEMM).  Replaces the unblocked
// interior of LAPACK potrf behind np.linalg.cholesky (GPr.py:62, GPpref.py:128).
//
// One workgroup of 1024 threads (16 waves) on one CU.
//  1. Elimination with the WHOLE tile in registers: thread (i, kg) owns row i and the column
//     pairs {16*q + 2*kg, +1 | q = 0..7} (cyclic pairs: balanced until the last columns, and
//     the published column is read back as 16-byte pairs).  Per column j the owners publish
//     column j and 1/pivot into a double-buffered LDS vector, ONE barrier, then every thread
//     applies the rank-1 update to its registers with no per-lane predicate: the full symmetric
//     tile is updated, the lower triangle is bit-for-bit what a masked update would give.
//     Square-root-free: column j and pivot j are final after step j, L = S*diag(1/sqrt(piv))
//     is applied once at the end.
//  2. In-place inverse of L in LDS (128 x 129 doubles, odd stride = conflict-free column
//     walks): the 16x16 diagonal blocks by per-wave forward substitution, then recursive doubling
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
#define CBLD 132  // published column: 128 values + 1/pivot at [128], padded to a multiple of 2

// Factor of the 16x16 diagonal block at column c0 of the LDS tile, executed by ONE wave, rank-4 updates on the
// matrix pipe.  The 16x16 block lives in ONE accumulator (C/D layout:
// lane (r = lane & 15, k = lane >> 4), register q holds element (row k + 4q, col r)) and stays fully symmetric, so
// register q of lane (r, k) IS the panel element (row r, column 4q + k) in MFMA operand layout: no data movement
// between "take the next four columns" and "apply them".  Per four columns: gather the panel row-wise (4
// ds_bpermute pairs), factor the four columns with readlane broadcasts (6 updates instead of 54), one
// v_mfma_f64_16x16x4_f64 for the whole trailing block.  ~100 instructions per four columns against ~240 for the
// lane-per-row version with 15 readlane broadcasts per column (measured 18.5 vs 19.4 us per leaf: the chain of
// dependent fp64 operations per pivot, not the instruction count, is what is left).
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
      const bool ok = (piv > 1.0e-300) && (piv < 1.0e300);
      if (!ok && !bad) bad = jb + c + 1;
      double y = __builtin_amdgcn_rsq(piv);
      const double hp = 0.5 * piv;
      y = __builtin_fma(y, __builtin_fma(-hp * y, y, 0.5), y);
      y = __builtin_fma(y, __builtin_fma(-hp * y, y, 0.5), y);
      double sq = piv * y;
      sq = __builtin_fma(0.5 * y, __builtin_fma(-sq, sq, piv), sq);
      l[c] = (r == jb + c) ? sq : ((r > jb + c) ? p[c] * y : 0.0);
      if (lane == 0) rinvb[jb + c] = y;
#pragma unroll
      for (int c2 = c + 1; c2 < 4; ++c2) {
        const double lkj = readlane_d(l[c], jb + c2);
        p[c2] = __builtin_fma(-l[c], lkj, p[c2]);
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

// The leaf's work on one tile.  S (128 * PLD + 2 * CBLD + 128 doubles, 16-byte aligned), rinvb (16) and failflag live
// in the calling kernel's LDS: the one-shot kernel below and the resident leaf server share this body.
// COH: the tile is read and the results are written THROUGH the caches (agent-scope accesses): the resident server
// neither invalidates nor writes back its XCD's L2 per tile, which the other kernels on that XCD would pay for.
template <bool COH> static __device__ __forceinline__ double leaf_ldg(const double* p) {
  return COH ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool COH> static __device__ __forceinline__ void leaf_stg(double* p, double v) {
  if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <int ALGO, bool COH = false>  // 0: column-by-column elimination in registers; 1: blocked elimination (default)
static __device__ __forceinline__ void leaf_body(const double* __restrict__ A, long lda, double* __restrict__ L, long ldl,
                                                 double* __restrict__ Dinv, int* info, int col0, int dbg_skip,
                                                 double* S, double* rinvb, int& failflag, const int tid) {
  constexpr int algo = ALGO;
  double* colbuf = S + 128 * PLD;       // [2][CBLD]; 128*129 and CBLD are even -> 16-byte aligned pair reads
  double* pivbuf = colbuf + 2 * CBLD;   // [128] pivots
  LEAF_STAMP(3);
  for (int e = tid; e < 128 * 128; e += 1024) {
    const int i = e >> 7, j = e & 127;
    S[i * PLD + j] = leaf_ldg<COH>(A + (long)i * lda + j);
  }
  __syncthreads();
  LEAF_STAMP(4);

  if (algo == 0) {
  const int i = tid & 127;
  const int kg = __builtin_amdgcn_readfirstlane(tid >> 7);  // wave-uniform -> scalar branches below
  double a[8][2];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    a[q][0] = S[i * PLD + 16 * q + 2 * kg];
    a[q][1] = S[i * PLD + 16 * q + 2 * kg + 1];
  }
  int fail = 0;
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int jq = 0; jq < 8; ++jq) {
    if (dbg_skip & 1) break;
#pragma nounroll
    for (int jkg = 0; jkg < 8; ++jkg) {  // NOT unrolled: the fully unrolled body overflows the instruction cache
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        const int j = 16 * jq + 2 * jkg + jh;
        double* cb = colbuf + (j & 1) * CBLD;
        if (kg == jkg) {
          const double v = a[jq][jh];
          cb[i] = v;
          if (i == j) {
            const bool ok = (v > 1.0e-300) && (v < 1.0e300);
            // reciprocal on the critical path of every column: v_rcp_f64 + two Newton steps (~1 ulp)
            double rcp = __builtin_amdgcn_rcp(v);
            rcp = rcp * (2.0 - v * rcp);
            rcp = rcp * (2.0 - v * rcp);
            cb[128] = ok ? rcp : -1.0;  // non-positive / NaN / Inf pivot -> flagged to everyone
            pivbuf[j] = v;
          }
        }
        __syncthreads();
        // issue every LDS read of this step back to back (one latency, not three), then test the pivot
        const double rinv = cb[128];
        const double ci = cb[i];
        d2 cq[8];
#pragma unroll
        for (int q = jq; q < 8; ++q) cq[q] = *reinterpret_cast<const d2*>(cb + 16 * q + 2 * kg);
        if (!(rinv > 0.0)) {  // block-uniform
          fail = j + 1;
          break;
        }
        const double lij = ci * rinv;
        // columns of this thread in block jq that are right of column j (wave-uniform tests)
        if (kg > jkg) {
          a[jq][0] -= lij * cq[jq].x;
          a[jq][1] -= lij * cq[jq].y;
        } else if (kg == jkg && jh == 0) {
          a[jq][1] -= lij * cq[jq].y;  // column j+1 is the second half of this thread's own pair
        }
#pragma unroll
        for (int q = jq + 1; q < 8; ++q) {
          a[q][0] -= lij * cq[q].x;
          a[q][1] -= lij * cq[q].y;
        }
      }
      if (fail) break;
    }
    if (fail) break;
  }
  __syncthreads();
  if ((dbg_skip & 8) && tid == 0) {  // diagnostic build path only: cycles / 100 MHz ticks of the elimination
    info[1] = (int)(__builtin_amdgcn_s_memtime() - st_c0);
    info[2] = (int)(__builtin_amdgcn_s_memrealtime() - st_r0);
  }
  if (fail) {
    if (tid == 0) atomicCAS(info, 0, col0 + fail);
    // keep downstream kernels on finite data: emit identity factors for this tile
    for (int e = tid; e < 128 * 128; e += 1024) {
      const int ii = e >> 7, jj = e & 127;
      const double v = (ii == jj) ? 1.0 : 0.0;
      leaf_stg<COH>(L + (long)ii * ldl + jj, v);
      leaf_stg<COH>(Dinv + ii * 128 + jj, v);
    }
    return;
  }
  // ---- scale: L[i][k] = a / sqrt(piv_k), L[k][k] = sqrt(piv_k), zero above; back to LDS
  double* isq = colbuf;  // [128] 1/sqrt(piv), computed once per column
  if (tid < 128) {
    const double dk = sqrt(pivbuf[tid]);
    pivbuf[tid] = dk;
    isq[tid] = 1.0 / dk;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 8; ++q) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 16 * q + 2 * kg + h;
      double x = 0.0;
      if (k < i) x = a[q][h] * isq[k];
      if (k == i) x = pivbuf[k];
      S[i * PLD + k] = x;
    }
  }
  __syncthreads();
  } else {
    // ---- blocked elimination (algo 1): 8 steps of 16 columns, the tile stays in LDS.
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
        leaf_stg<COH>(L + (long)ii * ldl + jj, v);
        leaf_stg<COH>(Dinv + ii * 128 + jj, v);
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
    leaf_stg<COH>(L + (long)ii * ldl + jj, S[ii * PLD + jj]);
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
    leaf_stg<COH>(Dinv + e, (jj <= ii) ? S[ii * PLD + jj] : 0.0);
  }
  LEAF_STAMP(7);
}


// ---------------------------------------------------------------- the slot-sized leaf (default)
// The same blocked algorithm -- operation for operation, so the factor and its inverse are BIT-IDENTICAL to leaf_body<1> --
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
#pragma unroll
  for (int I = 0; I < 8; ++I) {
    const int rl = 16 * (I + 1);
    for (int e = tid; e < 16 * rl; e += 512) {
      const int r = e / rl, c = e - r * rl;
      S[tri_base(I, c >> 4) + r * TLD + (c & 15)] = A[(long)(16 * I + r) * lda + c];
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
  for (int e = tid; e < 128 * 128; e += 512) {
    const int ii = e >> 7, jj = e & 127;
    L[(long)ii * ldl + jj] = ((jj >> 4) <= (ii >> 4)) ? S[tri_base(ii >> 4, jj >> 4) + (ii & 15) * TLD + (jj & 15)] : 0.0;
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
  for (int e = tid; e < 128 * 128; e += 512) {
    const int ii = e >> 7, jj = e & 127;
    Dinv[e] = ((jj >> 4) < (ii >> 4) || ((jj >> 4) == (ii >> 4) && jj <= ii)) ? S[tri_base(ii >> 4, jj >> 4) + (ii & 15) * TLD + (jj & 15)] : 0.0;
  }
  LEAF_STAMP(7);
}

__global__ __launch_bounds__(512, 4) void potf2_trtri_128_slot_kernel(const double* __restrict__ A, long lda, double* __restrict__ L,
                                                                      long ldl, double* __restrict__ Dinv, int* info, int col0,
                                                                      int dbg_skip) {
  __shared__ __attribute__((aligned(16))) double S[TRI_DOUBLES];
  __shared__ double rinvb[16];
  __shared__ int failflag;
  leaf_body_tri(A, lda, L, ldl, Dinv, info, col0, dbg_skip, S, rinvb, failflag, (int)threadIdx.x);
}

template <int ALGO>
__global__ __launch_bounds__(1024) void potf2_trtri_128_kernel(const double* __restrict__ A, long lda,
                                                               double* __restrict__ L, long ldl,
                                                               double* __restrict__ Dinv, int* info, int col0,
                                                               int dbg_skip) {
  __shared__ __attribute__((aligned(16))) double S[128 * PLD + 2 * CBLD + 128];
  __shared__ double rinvb[16];
  __shared__ int failflag;
  leaf_body<ALGO>(A, lda, L, ldl, Dinv, info, col0, dbg_skip, S, rinvb, failflag, (int)threadIdx.x);
}

int g_leaf_algo = 2;  // 2: slot-sized blocked leaf (512 threads, block-triangular LDS image; default); 1: the same algorithm, 1024 threads on the full tile; 0: column-by-column in registers
int launch_potf2_trtri_128(const double* A, long lda, double* L, long ldl, double* Dinv, int* info, int col0,
                           int dbg_skip, hipStream_t s) {
  if (g_leaf_algo == 2)
    hipLaunchKernelGGL(potf2_trtri_128_slot_kernel, dim3(1), dim3(512), 0, s, A, lda, L, ldl, Dinv, info, col0, dbg_skip);
  else if (g_leaf_algo == 0)
    hipLaunchKernelGGL(potf2_trtri_128_kernel<0>, dim3(1), dim3(1024), 0, s, A, lda, L, ldl, Dinv, info, col0, dbg_skip);
  else
    hipLaunchKernelGGL(potf2_trtri_128_kernel<1>, dim3(1), dim3(1024), 0, s, A, lda, L, ldl, Dinv, info, col0, dbg_skip);
  GPK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- resident leaf server (see gpk_internal.h)
static __device__ __forceinline__ unsigned mb_load(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ void mb_store(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int ALGO>
__global__ __launch_bounds__(1024) void leaf_server_kernel(LeafMailbox* mb, unsigned first_seq) {
  __shared__ __attribute__((aligned(16))) double S[128 * PLD + 2 * CBLD + 128];
  __shared__ double rinvb[16];
  __shared__ int failflag;
  __shared__ unsigned long long t_ptr[5];
  __shared__ long t_ld[2];
  __shared__ int t_int[3];  // col0, command (0 tile, 1 quit, 2 timed out)
  const int tid = threadIdx.x;
  for (unsigned seq = first_seq;; ++seq) {
    if (tid == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      int cmd = 0;
      while (mb_load(&mb->go) != seq) {
        __builtin_amdgcn_s_sleep(4);
        if (mb_load(&mb->err) != 0 || __builtin_amdgcn_s_memrealtime() - t0 > LEAF_SERVER_LIMIT_TICKS) {
          cmd = 2;
          break;
        }
      }
      if (cmd == 0) {
        const unsigned long long* q = reinterpret_cast<const unsigned long long*>(mb);
        t_ptr[0] = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // A
        t_ld[0] = (long)__hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t_ptr[1] = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L
        t_ld[1] = (long)__hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t_ptr[2] = __hip_atomic_load(q + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // Dinv
        t_ptr[3] = __hip_atomic_load(q + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // info
        const unsigned long long co = __hip_atomic_load(q + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t_int[0] = (int)(unsigned)(co & 0xffffffffull);
        cmd = (int)(unsigned)(co >> 32) ? 1 : 0;
      }
      t_int[1] = cmd;
    }
    __syncthreads();
    const int cmd = t_int[1];
    if (cmd == 2) {
      if (tid == 0) mb_store(&mb->err, 1u);
      return;
    }
    if (cmd == 1) {
      if (tid == 0) mb_store(&mb->done, seq);
      return;
    }
    // The tile was written by an earlier kernel of the chain (complete and written back before the proxy ran) and is
    // read past the caches; L and Dinv are written through them: no invalidate, no write-back of this XCD's L2.
    auto uni = [](unsigned long long v) -> unsigned long long {  // workgroup-uniform: keep it in scalar registers
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffull));
      const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
      return ((unsigned long long)hi << 32) | lo;
    };
    const double* tA = reinterpret_cast<const double*>(uni(t_ptr[0]));
    double* tL = reinterpret_cast<double*>(uni(t_ptr[1]));
    double* tD = reinterpret_cast<double*>(uni(t_ptr[2]));
    int* tinfo = reinterpret_cast<int*>(uni(t_ptr[3]));
    const long tlda = (long)uni((unsigned long long)t_ld[0]), tldl = (long)uni((unsigned long long)t_ld[1]);
    const int tcol0 = __builtin_amdgcn_readfirstlane(t_int[0]);
    int tid_it = tid;  // a fresh value per tile: the body's index arithmetic is NOT hoisted out of the server loop
    asm volatile("" : "+v"(tid_it));  // (hoisted, it stays live across the whole body: 11 spilled registers)
    leaf_body<ALGO, true>(tA, tlda, tL, tldl, tD, tinfo, tcol0, 0, S, rinvb, failflag, tid_it);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left before the barrier
    __syncthreads();
    if (tid == 0) mb_store(&mb->done, seq);
  }
}

__global__ __launch_bounds__(64) void leaf_proxy_kernel(LeafMailbox* mb, unsigned seq, int op, const double* A, long lda,
                                                        double* L, long ldl, double* Dinv, int* info, int col0) {
  if (threadIdx.x != 0) return;
  unsigned long long* q = reinterpret_cast<unsigned long long*>(mb);
  __hip_atomic_store(q + 0, (unsigned long long)A, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, (unsigned long long)lda, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 2, (unsigned long long)L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 3, (unsigned long long)ldl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 4, (unsigned long long)Dinv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 5, (unsigned long long)info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 6, (unsigned long long)(unsigned)col0 | ((unsigned long long)(unsigned)op << 32), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the task is in memory before `go` (earlier kernels' data: kernel boundary)
  mb_store(&mb->go, seq);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  bool ok = true;
  while (mb_load(&mb->done) != seq) {
    __builtin_amdgcn_s_sleep(2);
    if (mb_load(&mb->err) != 0) { ok = false; break; }
    if (__builtin_amdgcn_s_memrealtime() - t0 > LEAF_SERVER_LIMIT_TICKS) {
      mb_store(&mb->err, 2u);
      ok = false;
      break;
    }
  }
  if (!ok && info) atomicCAS(info, 0, LEAF_SERVER_INFO_POISON);
}

// The leaf of the FLAG-CHAINED tile pipeline: waits (bounded) for `wait_flag` -- set by the previous step's T kernel on the
// update stream; nullptr for the first tile -- factors its tile with cache-bypassing accesses (the tile was written by that
// still-running kernel through the caches), and sets `set_flag` when L and Dinv are out.
template <int ALGO>
__global__ __launch_bounds__(1024) void leaf_flag_kernel(const double* __restrict__ A, long lda, double* __restrict__ L, long ldl,
                                                         double* __restrict__ Dinv, int* info, int col0, const unsigned* wait_flag,
                                                         unsigned* set_flag) {
  __shared__ __attribute__((aligned(16))) double S[128 * PLD + 2 * CBLD + 128];
  __shared__ double rinvb[16];
  __shared__ int failflag;
  __shared__ int gave_up;
  const int tid = threadIdx.x;
  if (wait_flag) {
    if (tid == 0) {
      int bad = 0;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (mb_load(wait_flag) == 0u) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > LEAF_SERVER_LIMIT_TICKS) { bad = 1; break; }
      }
      gave_up = bad;
    }
    __syncthreads();
    if (gave_up && tid == 0 && info) atomicCAS(info, 0, LEAF_SERVER_INFO_POISON);
  }
  leaf_body<ALGO, true>(A, lda, L, ldl, Dinv, info, col0, 0, S, rinvb, failflag, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && set_flag) mb_store(set_flag, 1u);
}

int launch_leaf_flag(const double* A, long lda, double* L, long ldl, double* Dinv, int* info, int col0, const unsigned* wait_flag,
                     unsigned* set_flag, hipStream_t s) {
  if (g_leaf_algo == 0) hipLaunchKernelGGL(leaf_flag_kernel<0>, dim3(1), dim3(1024), 0, s, A, lda, L, ldl, Dinv, info, col0, wait_flag, set_flag);
  else hipLaunchKernelGGL(leaf_flag_kernel<1>, dim3(1), dim3(1024), 0, s, A, lda, L, ldl, Dinv, info, col0, wait_flag, set_flag);
  GPK_HIP(hipGetLastError());
  return 0;
}

int launch_leaf_server(LeafMailbox* mb, unsigned first_seq, hipStream_t s) {
  static_assert(offsetof(LeafMailbox, go) == 128 && offsetof(LeafMailbox, done) == 256 && offsetof(LeafMailbox, err) == 384,
                "mailbox layout");
  if (g_leaf_algo == 0) hipLaunchKernelGGL(leaf_server_kernel<0>, dim3(1), dim3(1024), 0, s, mb, first_seq);
  else hipLaunchKernelGGL(leaf_server_kernel<1>, dim3(1), dim3(1024), 0, s, mb, first_seq);
  GPK_HIP(hipGetLastError());
  return 0;
}

int launch_leaf_proxy(LeafMailbox* mb, unsigned seq, int op, const double* A, long lda, double* L, long ldl,
                      double* Dinv, int* info, int col0, hipStream_t s) {
  hipLaunchKernelGGL(leaf_proxy_kernel, dim3(1), dim3(64), 0, s, mb, seq, op, A, lda, L, ldl, Dinv, info, col0);
  GPK_HIP(hipGetLastError());
  return 0;
}
