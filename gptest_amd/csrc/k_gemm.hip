// fp64 MFMA GEMM, "NT" form:  C[m x n] (op)= A[m x k] * B[n x k]^T   (all row-major).
//
// This one kernel is the Cholesky trailing update (A = B = factored panel, lower
// tiles only: SYRK), the panel solve (B = inverse of the diagonal tile) and the
// forward-substitution update of the prediction solve.  It replaces the LAPACK
// potrf/getrf interior of np.linalg.cholesky / inv / solve at GPr.py:48,62-63.
//
// CDNA4 mapping
//   * v_mfma_f64_16x16x4_f64: 2048 flop / 64 cycles / SIMD = the fp64 peak
//     (78.6 TFLOP/s chip-wide).  One MFMA consumes ONE f64 of A and ONE of B per
//     lane, so a wave that register-blocks 4x4 MFMA tiles (64x64 outputs) needs
//     8 ds_read_b64 per 16 MFMAs (1024 cycles): LDS and L2 are nowhere near a
//     bound, the matrix pipe is.  4 waves (2x2) -> 128x128 block tile.
//   * A/B fragment: lane l holds M[r0 + (l&15)][k0 + (l>>4)]  (both operands are
//     k-contiguous, so A and B use the same LDS image and the same read).
//   * C/D fragment (f64 only!): col = lane&15, row = (lane>>4) + 4*reg.
//   * LDS row stride BK+2 doubles (register-staged forms): the 32 lanes of a ds_read_b64 group touch
//     dword banks 36*r + 2*g (+0,1) mod 64 -> all 64 banks once.  (The compiler merges pairs of those reads into
//     ds_read2_b64, which is banked mod 32 and half rate: measured 2-way conflicts, scripts/pmc_lds.sh.  The
//     LDS-DMA form issues its ds_read_b64 by inline asm and measures conflict-free.)
//   * 2 workgroups per CU (64 KiB of LDS each in the LDS-DMA form, 72 KiB register-staged) so one
//     workgroup's C tile hand-over overlaps the other's MFMA main loop.
#include "gpk_internal.h"
#include <type_traits>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define BK 16
#define LDS_LD (BK + 2)

// One block tile of the 4-wave kernel (block `bid` of the launch described by g); lds: 2 * 2 * BT * LDS_LD doubles.
// COH (the fused / flag-chained T step of the tile pipeline), a bit mask: 1 = C is stored THROUGH the caches (agent-scope
// stores), 2 = the A / B operands are read PAST them (agent-scope loads), so that concurrently running workgroups and
// kernels can hand a tile to each other without writing back or invalidating a whole L2.
template <int WT, int COH = 0>  // MFMA tiles per wave per dimension: 4 -> 128x128 block, 2 -> 64x64 block
static __device__ __forceinline__ void gemm_nt_body(const GemmNT& g, const int bid, double* lds) {
  constexpr int BT = 2 * WT * 16;       // block tile edge
  constexpr int PASSES = BT / 32;       // 32 rows of 16 doubles per pass with 256 threads
  double* As = lds;                     // [2][BT][LDS_LD]
  double* Bs = lds + 2 * BT * LDS_LD;   // [2][BT][LDS_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  // ---- tile decode
  int ti, tj;
  const int ntn = g.n / BT;
  if (g.lower && g.cyc_P == 0) {
    // lower trapezoid (m >= n, C(0,0) on the diagonal): the n x n triangle first, then the full rows below it
    int id = bid;
    const int tri = ntn * (ntn + 1) / 2;
    if (id < tri) {
      int t = (int)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
      while (t * (t + 1) / 2 > id) --t;
      while ((t + 1) * (t + 2) / 2 <= id) ++t;
      ti = t;
      tj = id - t * (t + 1) / 2;
    } else {
      id -= tri;
      ti = ntn + id / ntn;
      tj = id % ntn;
    }
  } else {
    ti = bid / ntn;
    tj = bid % ntn;
  }
  long brow0;  // first row of the B tile
  if (g.cyc_P > 0) {
    const int tpb = g.cyc_NB / BT;
    const int t = tj + g.cyc_c0 / BT;
    long gcol = ((long)(t / tpb + g.cyc_lg0) * g.cyc_P + g.cyc_r) * g.cyc_NB + (long)(t % tpb) * BT;
    if (g.lower && gcol > g.row_g0 + (long)ti * BT) return;  // tile entirely above the diagonal
    brow0 = gcol - g.brow_g0;
  } else {
    brow0 = (long)tj * BT;
  }
  const double* Ag = g.A + (long)ti * BT * g.lda;
  const double* Bg = g.B + brow0 * g.ldb;

  // ---- global -> register staging: 16-byte chunks, 8 chunks per 128-byte row segment
  const int lrow = tid >> 3;        // 0..31
  const int lchk = (tid & 7) * 2;   // double offset inside the BK row
  d2 ra[PASSES], rb[PASSES];
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const double* pa_ = Ag + (long)(p * 32 + lrow) * g.lda + kt * BK + lchk;
      const double* pb_ = Bg + (long)(p * 32 + lrow) * g.ldb + kt * BK + lchk;
      if (COH & 2) {
        ra[p] = (d2){__hip_atomic_load(pa_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(pa_ + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
        rb[p] = (d2){__hip_atomic_load(pb_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(pb_ + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
      } else {
        ra[p] = *reinterpret_cast<const d2*>(pa_);
        rb[p] = *reinterpret_cast<const d2*>(pb_);
      }
    }
  };
  // sub mode, exactly as in the 8-wave kernel below (so that every kernel of this file rounds identically and the
  // choice of tile size never changes a bit of the factor): accumulators start as C, A is negated on its way
  // into LDS, the epilogue only stores.
  const double sgn = g.sub ? -1.0 : 1.0;
  auto lstore = [&](int buf) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      d2 na = {sgn * ra[p].x, sgn * ra[p].y};
      *reinterpret_cast<d2*>(As + (buf * BT + p * 32 + lrow) * LDS_LD + lchk) = na;
      *reinterpret_cast<d2*>(Bs + (buf * BT + p * 32 + lrow) * LDS_LD + lchk) = rb[p];
    }
  };

  const int nk = g.k / BK;
  const int fr = lane & 15, fg = lane >> 4;
  double* Cg = g.C + ((long)ti * BT + wm * WT * 16) * g.ldc + (long)tj * BT + wn * WT * 16;
  d4 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      if (g.sub) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = Cg[(long)(i * 16 + fg + 4 * r) * g.ldc + j * 16 + fr];
      } else {
        acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
      }
    }
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const double* Ab = As + (buf * BT + wm * WT * 16 + fr) * LDS_LD + fg;
    const double* Bb = Bs + (buf * BT + wn * WT * 16 + fr) * LDS_LD + fg;
    if constexpr (WT == 2) {
      // one wave per SIMD and workgroup: the fragments of k-step s+1 are read under the MFMAs of k-step s (same MFMA
      // order, same bits), so that one LDS latency per slab is exposed instead of four
      double a[2][WT], b[2][WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) a[0][i] = Ab[i * 16 * LDS_LD];
#pragma unroll
      for (int j = 0; j < WT; ++j) b[0][j] = Bb[j * 16 * LDS_LD];
#pragma unroll
      for (int q = 0; q < BK / 4; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < BK / 4) {
#pragma unroll
          for (int i = 0; i < WT; ++i) a[(q + 1) & 1][i] = Ab[i * 16 * LDS_LD + 4 * (q + 1)];
#pragma unroll
          for (int j = 0; j < WT; ++j) b[(q + 1) & 1][j] = Bb[j * 16 * LDS_LD + 4 * (q + 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
          for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q & 1][i], b[q & 1][j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double a[WT], b[WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) a[i] = Ab[i * 16 * LDS_LD + kk];
#pragma unroll
      for (int j = 0; j < WT; ++j) b[j] = Bb[j * 16 * LDS_LD + kk];
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&15, row = (lane>>4) + 4*reg
#pragma unroll
  for (int i = 0; i < WT; ++i) {
#pragma unroll
    for (int j = 0; j < WT; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* pc_ = Cg + (long)(i * 16 + fg + 4 * r) * g.ldc + j * 16 + fr;
        if (COH & 1) __hip_atomic_store(pc_, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *pc_ = acc[i][j][r];
      }
    }
  }
}


template <int WT>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNT g) {
  __shared__ double lds[2 * 2 * (2 * WT * 16) * LDS_LD];
  gemm_nt_body<WT>(g, (int)blockIdx.x, lds);
}

// The tile pipeline's two one-tile GEMMs of a chain step in ONE launch of four 64x64-tile workgroups: block b first does
// block b of T1 (L21 = A21 Dinv^T, four blocks), publishes it (stores through the caches, one count), and -- blocks 0..2 -- waits
// until all four are counted before doing block b of T2 (the three lower blocks of A22 -= L21 L21^T).  Same tile code
// as two launches, so the same bits; one dependent launch less on the chain.  `ctr` counts up over the steps of one
// factorisation (launches are stream-ordered, so a step's blocks never see the next step's counts); the wait is bounded
// (5 s of the 100 MHz clock), a time-out poisons *info like the leaf server does.
__global__ __launch_bounds__(256, 2) void tt_fused_kernel(GemmNT t1, GemmNT t2, unsigned* ctr, unsigned target, int* info) {
  __shared__ double lds[2 * 2 * 64 * LDS_LD];
  __shared__ int gave_up;
  const int b = (int)blockIdx.x;
  gemm_nt_body<2, 1>(t1, b, lds);  // L21 written through the caches
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... and out of this wave before the block is counted
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int bad = 0;
    if (b < 3) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > LEAF_SERVER_LIMIT_TICKS) { bad = 1; break; }
      }
    }
    gave_up = bad;
  }
  __syncthreads();
  if (b >= 3) return;
  if (gave_up) {
    if (threadIdx.x == 0 && info) atomicCAS(info, 0, LEAF_SERVER_INFO_POISON);
    return;
  }
  gemm_nt_body<2, 2>(t2, b, lds);  // the other blocks' L21 is read past the caches (this XCD's L2 may hold last posterior's)
}

int g_gemm_variant = 8;  // 8: 8-wave 128x128 kernel (default); 4: the 4-wave kernel
int g_gemm_pipe = 2;     // 8-wave kernel: 2 software-pipelined slab loop staged by LDS-DMA (default), 1 the same with register staging, 0 the plain loop

// ---------------------------------------------------------------------------------------------
// 8-wave variant of the 128x128 tile: wave tile 64x32 (4x2 MFMA tiles, 64 accumulator VGPRs), <=128
// VGPRs per wave, two workgroups per CU = FOUR waves per SIMD in two independent barrier domains.  The matrix
// pipe itself is saturated by ONE wave per SIMD (gpk_bench_mfma_pinned: 98-99 % of 2048 flop / 64 cycles / SIMD
// at W = 1, 2, 4, 8; the n/(n+1) figures of gpk_bench_mfma_peak2 are workgroup placement); what the second
// workgroup buys is cover for the first one's barrier, C tile and turnover (one workgroup per CU: 62 TFLOP/s,
// two: 66-69).
// sub mode: the accumulators START as C and A enters the MFMA negated (in the DMA form by the instruction's
// neg bit, in the register-staged forms on its way into LDS), so the MFMA chain itself yields C - A*B^T and
// the epilogue is store-only (no load -> subtract -> store dependency at the end of every tile).
// Three forms of the slab loop, bitwise the same C (option gemm_pipe): 2 LDS-DMA staging + fragment prefetch
// (default), 1 register staging + fragment prefetch, 0 the plain loop.  DESIGN.md section 6 has the numbers.
// one wave-instruction of LDS-DMA: every lane's 16 bytes at gsrc go to LDS byte address lds_dst + 16 * lane (lds_dst wave-uniform);
// no VGPR destination, counted on vmcnt, invisible to the compiler's own wait bookkeeping
// (source = wave-uniform base + per-lane 32-bit byte offset: one address VGPR per operand for the whole kernel)
static __device__ __forceinline__ void glds16(const double* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

template <bool TWOSEG, int DBG = 0, bool PIPE = false, bool GLDS = false, bool NEG = false>  // TWOSEG: the K range continues in (A2, B2) after k1 columns (paired trailing update)
__global__ __launch_bounds__(512, 4) void gemm_nt8_kernel(GemmNT g) {
  constexpr int BT = 128;
  // LDS-DMA form: the unpadded image, exactly 64 KiB -- two workgroups per CU leave 32 KiB, and ONE finished workgroup
  // frees a 96 KiB slot, which is what the slot-sized leaf (k_potf2.hip) is built to fit
  __shared__ double lds[2 * 2 * BT * (GLDS ? BK : LDS_LD)];
  double* As = lds;
  double* Bs = lds + 2 * BT * LDS_LD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;  // 2 x 4 waves of 64 x 32

  int ti, tj;
  const int ntn = g.n / BT;
  if (g.xcd_sq && g.cyc_P == 0) {
    // XCD-aware order: blocks b and b+8 share an XCD (observed, speed only).  The tile grid is cut into
    // 8x8-tile squares; the 64 consecutive blocks of one XCD take one square, so the tiles an XCD works on
    // share 8 A and 8 B row-blocks (4 MiB at k = 256: its L2) instead of streaming a new B block per tile.
    const int ntm = g.m / BT;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int sq = (seq >> 6) * 8 + xcd, slot = seq & 63;
    const int SN = (ntn + 7) >> 3;
    int I, J;
    if (g.lower && (g.xcd_sq & 4)) {
      // sorted enumeration: the full squares first (the rows below the triangle, then the squares strictly below the
      // diagonal), the half-empty diagonal squares last, so that the eight squares of a round (one per XCD) carry the
      // same number of real tiles: workgroups are dispatched in index order, and an XCD that runs out of real work
      // in its square while the next index belongs to a full XCD stalls everyone behind it.
      const int SM = (ntm + 7) >> 3;
      const int R = (SM - SN) * SN, F = SN * (SN - 1) / 2;
      if (sq < R) {
        I = SN + sq / SN;
        J = sq % SN;
      } else if (sq - R < F) {
        const int q = sq - R;
        int r = (int)((sqrtf(8.0f * (float)q + 1.0f) + 1.0f) * 0.5f);
        while (r * (r - 1) / 2 > q) --r;
        while ((r + 1) * r / 2 <= q) ++r;
        I = r;
        J = q - r * (r - 1) / 2;
      } else {
        I = J = sq - R - F;
      }
    } else if (g.lower) {  // lower trapezoid of squares: SN x SN triangle, then full rows of SN squares
      const int tri = SN * (SN + 1) / 2;
      if (sq < tri) {
        int r = (int)((sqrtf(8.0f * (float)sq + 1.0f) - 1.0f) * 0.5f);
        while (r * (r + 1) / 2 > sq) --r;
        while ((r + 1) * (r + 2) / 2 <= sq) ++r;
        I = r;
        J = sq - r * (r + 1) / 2;
      } else {
        I = SN + (sq - tri) / SN;
        J = (sq - tri) % SN;
      }
    } else {
      I = sq / SN;
      J = sq - I * SN;
    }
    ti = I * 8 + (slot >> 3);
    tj = J * 8 + (slot & 7);
    if (ti >= ntm || tj >= ntn || (g.lower && tj > ti)) return;
  } else if (g.lower && g.cyc_P == 0) {
    int id = blockIdx.x;
    const int tri = ntn * (ntn + 1) / 2;
    if (id < tri) {
      int t = (int)((sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
      while (t * (t + 1) / 2 > id) --t;
      while ((t + 1) * (t + 2) / 2 <= id) ++t;
      ti = t;
      tj = id - t * (t + 1) / 2;
    } else {
      id -= tri;
      ti = ntn + id / ntn;
      tj = id % ntn;
    }
  } else {
    ti = blockIdx.x / ntn;
    tj = blockIdx.x % ntn;
  }
  long brow0;
  if (g.cyc_P > 0) {
    const int tpb = g.cyc_NB / BT;
    const int t = tj + g.cyc_c0 / BT;
    long gcol = ((long)(t / tpb + g.cyc_lg0) * g.cyc_P + g.cyc_r) * g.cyc_NB + (long)(t % tpb) * BT;
    if (g.lower && gcol > g.row_g0 + (long)ti * BT) return;
    brow0 = gcol - g.brow_g0;
  } else {
    brow0 = (long)tj * BT;
  }
  const double* Ag = (DBG & 16) ? g.A : g.A + (long)ti * BT * g.lda;  // DBG 16: every tile reads tile 0's first slab (cache hits)
  const double* Bg = (DBG & 16) ? g.B : g.B + brow0 * g.ldb;
  const int lrow = tid >> 3, lchk = (tid & 7) * 2;  // 64 rows x 8 chunks per pass
  const int fr = lane & 15, fg = lane >> 4;
  double* Cw = g.C + ((long)ti * BT + wm * 64) * g.ldc + (long)tj * BT + wn * 32;  // wave-uniform
  const unsigned c_off = (unsigned)fg * (unsigned)g.ldc + (unsigned)fr;

  d4 acc[4][2];
  auto init_acc = [&]() {
  if (g.sub && !(DBG & 8)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)  // read once per launch: non-temporal, so that the C stream does not evict panel lines from L2
          acc[i][j][r] = __builtin_nontemporal_load((Cw + (long)(i * 16 + 4 * r) * g.ldc + j * 16) + c_off);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  }
  };
  if (!GLDS) init_acc();  // (LDS-DMA staging: the first two slabs are requested BEFORE the C tile, which queues behind them in the load path)

  d2 ra[2], rb[2];
  const int nk1 = TWOSEG ? g.k1 / BK : 0;  // slabs of the first K segment
  const double* Ag2 = TWOSEG ? g.A2 + (long)ti * BT * g.lda2 : Ag;
  const double* Bg2 = TWOSEG ? g.B2 + brow0 * g.ldb2 : Bg;
  auto gload = [&](int kt) {
    if (DBG & 16) kt = 0;
    if (TWOSEG) {
      // wave-uniform segment select: slabs [0, nk1) from (A, B), the rest from (A2, B2).  Both segments have the
      // same leading dimensions (host check), so only the scalar base and slab index differ from the plain path:
      // a per-slab select of the leading dimension cost 5 % of the kernel (per-thread 64-bit offset arithmetic).
      const bool first = kt < nk1;
      const double* Ap = first ? Ag : Ag2;
      const double* Bp = first ? Bg : Bg2;
      const int kk = (first ? kt : kt - nk1) * BK + lchk;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        ra[p] = *reinterpret_cast<const d2*>(Ap + (long)(p * 64 + lrow) * g.lda + kk);
        rb[p] = *reinterpret_cast<const d2*>(Bp + (long)(p * 64 + lrow) * g.ldb + kk);
      }
    } else {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        ra[p] = *reinterpret_cast<const d2*>(Ag + (long)(p * 64 + lrow) * g.lda + kt * BK + lchk);
        rb[p] = *reinterpret_cast<const d2*>(Bg + (long)(p * 64 + lrow) * g.ldb + kt * BK + lchk);
      }
    }
  };
  const double sgn = g.sub ? -1.0 : 1.0;
  auto lstore = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      d2 na = {sgn * ra[p].x, sgn * ra[p].y};
      *reinterpret_cast<d2*>(As + (buf * BT + p * 64 + lrow) * LDS_LD + lchk) = na;
      *reinterpret_cast<d2*>(Bs + (buf * BT + p * 64 + lrow) * LDS_LD + lchk) = rb[p];
    }
  };

  const int nk = g.k / BK;
  if constexpr (GLDS) {
    // Staging by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write, no negation pass (the MFMA
    // negates A: BLGP bit 0 of the f64 MFMAs).  The LDS image is lane-linear per wave-instruction -- 8 rows of 128 bytes
    // -- so rows are unpadded and the bank conflicts are swizzled away on the SOURCE side: LDS (row r, 16-byte chunk c)
    // holds global chunk c ^ ((r >> 1) & 7) of that row's slab, and the fragment reads apply the same XOR: the 32 lanes
    // of a ds_read_b64 group (16 rows x 2 halves of one chunk column) then cover all 64 banks once.
    constexpr int OPD = BT * BK;   // doubles per operand and stage (128 rows x 16)
    constexpr int STG = 2 * OPD;   // A then B: 32 KiB per stage, two stages
    typedef __attribute__((address_space(3))) double lds_double;
    const unsigned lds0 = (unsigned)(unsigned long)(lds_double*)lds;
    const int rr = wid * 8 + (lane >> 3);  // the row this lane fills (and rr + 64)
    const int gch = ((lane & 7) ^ ((rr >> 1) & 7)) * 2;
    const unsigned voA = (unsigned)(((long)rr * g.lda + gch) * 8);  // < 4 GiB: 128 rows of one operand (host check)
    const unsigned voB = (unsigned)(((long)rr * g.ldb + gch) * 8);
    const long a64 = 64 * g.lda, b64 = 64 * g.ldb;
    const int nk1g = TWOSEG ? g.k1 / BK : 0;
    const double* Ag2g = TWOSEG ? g.A2 + (long)ti * BT * g.lda2 : Ag;  // second K segment: same leading dimensions (host check)
    const double* Bg2g = TWOSEG ? g.B2 + brow0 * g.ldb2 : Bg;
    auto dma = [&](int kt, int buf) {
      const unsigned dst = lds0 + (unsigned)((buf * STG + wid * 8 * BK) * 8);
      const bool first = !TWOSEG || kt < nk1g;  // scalar select of the segment: slabs [0, nk1) from (A, B), the rest from (A2, B2)
      const double* Ap = (first ? Ag : Ag2g) + (first ? kt : kt - nk1g) * BK;
      const double* Bp = (first ? Bg : Bg2g) + (first ? kt : kt - nk1g) * BK;
      glds16(Ap, voA, dst);
      glds16(Ap + a64, voA, dst + 64 * BK * 8);
      glds16(Bp, voB, dst + OPD * 8);
      glds16(Bp + b64, voB, dst + OPD * 8 + 64 * BK * 8);
    };
    const int fsw = (fr >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = 4 * q + fg;
      koff[q] = ((k >> 1) ^ fsw) * 2 + (k & 1);
    }
    // Fragment reads as single ds_read_b64 by inline asm: left to the compiler, pairs of them are merged into
    // ds_read2st64_b64, which is banked mod 32 in 16-lane groups and half rate -- 16 LDS-array cycles per instruction with
    // a 2-way conflict under this swizzle instead of 2 conflict-free cycles per ds_read_b64 (scripts/lds_bank_probe.sh,
    // scripts/pmc_lds.sh: SQ_LDS_BANK_CONFLICT 50 % of SQ_LDS_IDX_ACTIVE before, 0 after).  hipcc does not count asm
    // loads, so the waits are explicit: LDS returns in order, `lgkmcnt(6)` retires the older of two fragment sets.
    unsigned adA[4], adB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      adA[q] = lds0 + (unsigned)(((wm * 64 + fr) * BK + koff[q]) * 8);
      adB[q] = lds0 + (unsigned)((OPD + (wn * 32 + fr) * BK + koff[q]) * 8);
    }
    double pa[2][4], pb[2][2];
#define GPK_DSR(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    auto rd = [&](int slot, int buf, int q) {
      const unsigned a = adA[q] + (unsigned)buf * (STG * 8), b = adB[q] + (unsigned)buf * (STG * 8);
      GPK_DSR(pa[slot][0], a, 0 * 16 * BK * 8);
      GPK_DSR(pa[slot][1], a, 1 * 16 * BK * 8);
      GPK_DSR(pa[slot][2], a, 2 * 16 * BK * 8);
      GPK_DSR(pa[slot][3], a, 3 * 16 * BK * 8);
      GPK_DSR(pb[slot][0], b, 0 * 16 * BK * 8);
      GPK_DSR(pb[slot][1], b, 1 * 16 * BK * 8);
    };
#undef GPK_DSR
    // the fragment set `slot` is in registers once at most `newer` younger LDS reads are outstanding
    auto landed6 = [&](int slot) {
      asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(pa[slot][0]), "+v"(pa[slot][1]), "+v"(pa[slot][2]), "+v"(pa[slot][3]), "+v"(pb[slot][0]), "+v"(pb[slot][1]));
    };
    auto landed0 = [&](int slot) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pa[slot][0]), "+v"(pa[slot][1]), "+v"(pa[slot][2]), "+v"(pa[slot][3]), "+v"(pb[slot][0]), "+v"(pb[slot][1]));
    };
    {
      constexpr int NEGA = NEG ? 1 : 0;  // the host instantiates NEG = (sub != 0)
      auto mm = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[slot][i], pb[slot][j], acc[i][j], 0, 0, NEGA);
      };
      dma(0, 0);
      if (nk > 1) dma(1, 1);
      init_acc();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      rd(0, 0, 0);
      for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        __builtin_amdgcn_sched_barrier(0);
        rd(1, buf, 1);
        landed6(0);
        __builtin_amdgcn_sched_barrier(0);
        mm(0);
        __builtin_amdgcn_sched_barrier(0);
        rd(0, buf, 2);
        landed6(1);
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
        __builtin_amdgcn_sched_barrier(0);
        rd(1, buf, 3);
        landed6(0);
        __builtin_amdgcn_sched_barrier(0);
        mm(0);
        __builtin_amdgcn_sched_barrier(0);
        // the last fragments of this slab IN REGISTERS before the barrier releases its buffer
        landed0(1);
        // B_kt: every wave's DMA of slab kt+1 (issued a slab ago) has landed, this slab's buffer is released
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(DBG & 2)) __syncthreads();
        if (kt + 1 < nk) rd(0, buf ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(DBG & 1) && kt + 2 < nk) dma(kt + 2, buf);
      }
    }
  } else {
  gload(0);
  lstore(0);
  __syncthreads();
  if constexpr (PIPE) {
    // Software-pipelined slab loop: the fragments of k-step s+1 are read from LDS while the MFMAs of k-step s run
    // (two fragment sets, the same 24 registers the plain loop holds for two k-steps at once), the next slab goes to
    // LDS before the LAST k-step and the barrier sits in front of that k-step, so that the first fragments of the
    // next slab are read under its MFMAs.  Same MFMA order over k: bitwise the plain loop's result.
    double pa[2][4], pb[2][2];
    const double* Ab0 = As + (wm * 64 + fr) * LDS_LD + fg;
    const double* Bb0 = Bs + (wn * 32 + fr) * LDS_LD + fg;
    auto rd = [&](int slot, int buf, int kk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) pa[slot][i] = Ab0[(buf * BT + i * 16) * LDS_LD + kk];
#pragma unroll
      for (int j = 0; j < 2; ++j) pb[slot][j] = Bb0[(buf * BT + j * 16) * LDS_LD + kk];
    };
    auto mm = [&](int slot) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[slot][i], pb[slot][j], acc[i][j], 0, 0, 0);
    };
    // Staging (one register set): the slab written to LDS right after barrier B_kt is slab kt+2, into the buffer the
    // barrier has just released; its global loads were issued one slab earlier, and the registers are refilled with
    // slab kt+3 at once.  So a slab is in LDS a whole slab before it is read: no barrier ever waits on a fresh
    // ds_write, and the global loads have a full slab to land.
    if (nk > 1) gload(1);
    rd(0, 0, 0);
    if (nk > 1) {
      lstore(1);
      if (nk > 2) gload(2);
    }
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      __builtin_amdgcn_sched_barrier(0);
      rd(1, buf, 4);
      __builtin_amdgcn_sched_barrier(0);
      mm(0);
      __builtin_amdgcn_sched_barrier(0);
      rd(0, buf, 8);
      __builtin_amdgcn_sched_barrier(0);
      mm(1);
      __builtin_amdgcn_sched_barrier(0);
      rd(1, buf, 12);
      __builtin_amdgcn_sched_barrier(0);
      mm(0);
      __builtin_amdgcn_sched_barrier(0);
      // the last fragments of this slab must be IN REGISTERS before the barrier releases its buffer; left alone the
      // compiler sinks those ds_reads below the barrier (legal for a fence, fatal here: the buffer is rewritten behind it)
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(pa[1][i]));
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(pb[1][j]));
      if (!(DBG & 2)) __syncthreads();  // B_kt: slab kt+1 (written a slab ago) visible, this slab's buffer released
      if (kt + 1 < nk) rd(0, buf ^ 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm(1);
      __builtin_amdgcn_sched_barrier(0);
      if (!(DBG & 1) && kt + 2 < nk) {
        lstore(buf);
        if (kt + 3 < nk) gload(kt + 3);
      }
    }
  } else {
  double fa[4], fb[2];
  if (DBG & 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = As[(wm * 64 + fr + i * 16) * LDS_LD + fg];
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = Bs[(wn * 32 + fr + j * 16) * LDS_LD + fg];
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (!(DBG & 1) && kt + 1 < nk) gload(kt + 1);
    const double* Ab = As + (buf * BT + wm * 64 + fr) * LDS_LD + fg;
    const double* Bb = Bs + (buf * BT + wn * 32 + fr) * LDS_LD + fg;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = (DBG & 4) ? fa[i] : Ab[i * 16 * LDS_LD + kk];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = (DBG & 4) ? fb[j] : Bb[j * 16 * LDS_LD + kk];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (!(DBG & 1) && kt + 1 < nk) lstore(buf ^ 1);
    if (!(DBG & 2)) __syncthreads();
  }
  }
  }
  if (DBG & 8) {
    double t = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) t += acc[i][j][r];
    if (t == 12345.678) Cw[c_off] = t;
    return;
  }
  // Write-through (sc1) stores: the C tiles are not re-read before the next panel step, and lines left dirty in
  // the XCD L2s make EVERY kernel boundary on the chip pay for their write-back (the end-of-kernel release of the
  // latency-critical leaf on its reserved CU went from 60 us alone to 155-173 us beside this kernel).
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        __hip_atomic_store((Cw + (long)(i * 16 + 4 * r) * g.ldc + j * 16) + c_off, acc[i][j][r], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// The T step of the FLAG-CHAINED tile pipeline (potrf_tail, option tail_flags): launched on the update stream AHEAD of
// the leaf it depends on, its four workgroups spin (bounded) until the leaf kernel on the chain stream has set
// `leaf_flag`, do T1 (operands past the caches: Dinv comes from a kernel that is still running elsewhere; result through
// them), count, do T2 (blocks 0..2; operands past, result through the caches: the next leaf reads it past them), and the
// last T2 block to finish sets `tt_flag`, which is what the next leaf kernel waits for.  No event packet and no dependent
// launch is left on the chain: ~58 instead of ~85 us per 128 columns.  ctr[0] counts T1 blocks, ctr[1] T2 blocks.
__global__ __launch_bounds__(256, 2) void tt_flag_kernel(GemmNT t1, GemmNT t2, const unsigned* leaf_flag, unsigned* ctr, unsigned* tt_flag,
                                                         int* info) {
  __shared__ double lds[2 * 2 * 64 * LDS_LD];
  __shared__ int gave_up;
  const int b = (int)blockIdx.x;
  if (threadIdx.x == 0) {
    int bad = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(leaf_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
      __builtin_amdgcn_s_sleep(2);
      if (__builtin_amdgcn_s_memrealtime() - t0 > LEAF_SERVER_LIMIT_TICKS) { bad = 1; break; }
    }
    gave_up = bad;
  }
  __syncthreads();
  if (gave_up && threadIdx.x == 0 && info) atomicCAS(info, 0, LEAF_SERVER_INFO_POISON);  // (and go on: the followers must not wait as well)
  gemm_nt_body<2, 3>(t1, b, lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int bad = 0;
    if (b < 3) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 4u) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > LEAF_SERVER_LIMIT_TICKS) { bad = 1; break; }
      }
    }
    gave_up = bad;
  }
  __syncthreads();
  if (b >= 3) return;
  if (gave_up && threadIdx.x == 0 && info) atomicCAS(info, 0, LEAF_SERVER_INFO_POISON);
  gemm_nt_body<2, 3>(t2, b, lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2u)
    __hip_atomic_store(tt_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int launch_tt_flag(const GemmNT& t1, const GemmNT& t2, const unsigned* leaf_flag, unsigned* ctr2, unsigned* tt_flag, int* info, hipStream_t s) {
  if (t1.m != 128 || t1.n != 128 || t2.m != 128 || t2.n != 128 || t1.k % BK || t2.k % BK || t1.lower || !t2.lower || t1.cyc_P || t2.cyc_P ||
      t1.k1 || t2.k1 || !leaf_flag || !ctr2 || !tt_flag)
    return gpk_set_err(-1, "tt_flag: one 128x128 tile each, T1 full, T2 lower, three flag words");
  hipLaunchKernelGGL(tt_flag_kernel, dim3(4), dim3(256), 0, s, t1, t2, leaf_flag, ctr2, tt_flag, info);
  GPK_HIP(hipGetLastError());
  return 0;
}

int launch_tt_fused(const GemmNT& t1, const GemmNT& t2, unsigned* ctr, unsigned target, int* info, hipStream_t s) {
  if (t1.m != 128 || t1.n != 128 || t2.m != 128 || t2.n != 128 || t1.k % BK || t2.k % BK || t1.lower || !t2.lower || t1.cyc_P || t2.cyc_P ||
      t1.k1 || t2.k1)
    return gpk_set_err(-1, "tt_fused: one 128x128 tile each, T1 full, T2 lower");
  hipLaunchKernelGGL(tt_fused_kernel, dim3(4), dim3(256), 0, s, t1, t2, ctr, target, info);
  GPK_HIP(hipGetLastError());
  return 0;
}

int launch_gemm_nt(const GemmNT& g, bool tile128, hipStream_t s) {
  const int BT = tile128 ? 128 : 64;
  if (g.m <= 0 || g.n <= 0) return 0;
  if (g.m % BT || g.n % BT || g.k % BK || g.k <= 0)
    return gpk_set_err(-1, "gemm_nt: bad shape m=%d n=%d k=%d (tile %d)", g.m, g.n, g.k, BT);
  if ((g.lda & 1) || (g.ldb & 1) || ((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15))
    return gpk_set_err(-1, "gemm_nt: operands must be 16-byte aligned with even leading dimensions");
  if (g.lda >= (1l << 22) || g.ldb >= (1l << 22))  // LDS-DMA staging: 128 rows of an operand within a 32-bit byte offset
    return gpk_set_err(-1, "gemm_nt: leading dimensions must stay below 2^22");
  if (g.k1 > 0) {
    if (!tile128 || g_gemm_variant != 8 || g.cyc_P != 0)
      return gpk_set_err(-1, "gemm_nt: a second K segment needs the 8-wave single-GPU kernel");
    if (g.k1 % BK || g.k1 >= g.k || !g.A2 || !g.B2 || g.lda2 != g.lda || g.ldb2 != g.ldb || ((uintptr_t)g.A2 & 15) || ((uintptr_t)g.B2 & 15))
      return gpk_set_err(-1, "gemm_nt: bad second K segment (k1=%d k=%d)", g.k1, g.k);
  }
  const int ntm = g.m / BT, ntn = g.n / BT;
  long nblk;
  if (g.lower && g.cyc_P == 0) {
    if (g.m < g.n) return gpk_set_err(-1, "gemm_nt: lower needs m >= n (m=%d n=%d)", g.m, g.n);
    nblk = (long)ntn * (ntn + 1) / 2 + (long)(ntm - ntn) * ntn;
  } else {
    nblk = (long)ntm * ntn;
  }
  if (tile128 && g_gemm_variant == 8) {
    if (g.xcd_sq && g.cyc_P == 0) {  // whole 8x8 squares, 8 squares (one per XCD) per round
      const long SM = (ntm + 7) / 8, SN = (ntn + 7) / 8;
      const long nsq = g.lower ? SN * (SN + 1) / 2 + (SM - SN) * SN : SM * SN;
      nblk = ((nsq + 7) / 8) * 8 * 64;
    }
    // g_gemm_pipe (option "gemm_pipe", default 1): the software-pipelined slab loop; 0 = the plain loop (bitwise the same C).
    const dim3 grid((unsigned)nblk), block(512);
#ifdef GPK_GEMM_DBG_VARIANTS
    // timing-only builds of the kernel (make DBG=1; scripts/gemm_dbg.py): GPK_GEMM_DBG bits 1 no global loads / LDS
    // writes in the loop, 2 no barriers, 4 no LDS reads (plain loop only), 8 no C load / store, 16 every tile reads tile
    // 0's first slab; GPK_GEMM_DYNLDS=<bytes> of extra LDS per workgroup (20480: one workgroup per CU).  Wrong results.
    static const int dbg = getenv("GPK_GEMM_DBG") ? atoi(getenv("GPK_GEMM_DBG")) : 0;
    static const int dyn = getenv("GPK_GEMM_DYNLDS") ? atoi(getenv("GPK_GEMM_DYNLDS")) : 0;
    if ((dbg || dyn) && g.k1 == 0) {
#define GPK_DBG_LAUNCH(D, P)                                                                                              \
  do {                                                                                                                    \
    if (dyn > 0) (void)hipFuncSetAttribute((const void*)gemm_nt8_kernel<false, D, P>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn); \
    hipLaunchKernelGGL((gemm_nt8_kernel<false, D, P>), grid, block, dyn, s, g);                                           \
  } while (0)
      if (g_gemm_pipe == 2 && g.sub) {
#define GPK_DBG_LAUNCH_G(D)                                                                                               \
  do {                                                                                                                    \
    if (dyn > 0) (void)hipFuncSetAttribute((const void*)gemm_nt8_kernel<false, D, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn); \
    hipLaunchKernelGGL((gemm_nt8_kernel<false, D, true, true, true>), grid, block, dyn, s, g);                            \
  } while (0)
        switch (dbg) {
          case 1: GPK_DBG_LAUNCH_G(1); break;
          case 2: GPK_DBG_LAUNCH_G(2); break;
          case 8: GPK_DBG_LAUNCH_G(8); break;
          case 9: GPK_DBG_LAUNCH_G(9); break;
          case 16: GPK_DBG_LAUNCH_G(16); break;
          default: GPK_DBG_LAUNCH_G(0); break;
        }
#undef GPK_DBG_LAUNCH_G
      } else if (g_gemm_pipe) {
        switch (dbg) {
          case 1: GPK_DBG_LAUNCH(1, true); break;
          case 2: GPK_DBG_LAUNCH(2, true); break;
          case 3: GPK_DBG_LAUNCH(3, true); break;
          case 8: GPK_DBG_LAUNCH(8, true); break;
          case 11: GPK_DBG_LAUNCH(11, true); break;
          case 16: GPK_DBG_LAUNCH(16, true); break;
          default: GPK_DBG_LAUNCH(0, true); break;
        }
      } else {
        switch (dbg) {
          case 1: GPK_DBG_LAUNCH(1, false); break;
          case 2: GPK_DBG_LAUNCH(2, false); break;
          case 4: GPK_DBG_LAUNCH(4, false); break;
          case 7: GPK_DBG_LAUNCH(7, false); break;
          case 8: GPK_DBG_LAUNCH(8, false); break;
          case 15: GPK_DBG_LAUNCH(15, false); break;
          default: GPK_DBG_LAUNCH(0, false); break;
        }
      }
#undef GPK_DBG_LAUNCH
    } else
#endif
    if (g.k1 > 0 && g_gemm_pipe == 2 && g.sub) hipLaunchKernelGGL((gemm_nt8_kernel<true, 0, true, true, true>), grid, block, 0, s, g);
    else if (g.k1 > 0 && g_gemm_pipe == 2) hipLaunchKernelGGL((gemm_nt8_kernel<true, 0, true, true, false>), grid, block, 0, s, g);
    else if (g.k1 > 0) hipLaunchKernelGGL(gemm_nt8_kernel<true>, grid, block, 0, s, g);
    else if (g_gemm_pipe == 2 && g.sub) hipLaunchKernelGGL((gemm_nt8_kernel<false, 0, true, true, true>), grid, block, 0, s, g);
    else if (g_gemm_pipe == 2) hipLaunchKernelGGL((gemm_nt8_kernel<false, 0, true, true, false>), grid, block, 0, s, g);
    else if (g_gemm_pipe) hipLaunchKernelGGL((gemm_nt8_kernel<false, 0, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL(gemm_nt8_kernel<false>, grid, block, 0, s, g);
  }
  else if (tile128)
    hipLaunchKernelGGL(gemm_nt_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL(gemm_nt_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, s, g);
  GPK_HIP(hipGetLastError());
  return 0;
}
