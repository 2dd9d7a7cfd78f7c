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
//   * 64x64-tile kernel (register-staged): LDS row stride BK+2 doubles, the 32 lanes of a ds_read_b64 group touch
//     dword banks 36*r + 2*g (+0,1) mod 64 -> all 64 banks once.  (The compiler merges pairs of those reads into
//     ds_read2_b64, which is banked mod 32 and half rate: measured 2-way conflicts, scripts/pmc_lds.sh.)
//   * 8-wave 128x128-tile kernel (LDS-DMA staged): unpadded source-swizzled image, ds_read_b64 by inline asm,
//     measured conflict-free; 2 workgroups per CU (64 KiB of LDS each) so one workgroup's C tile hand-over
//     overlaps the other's MFMA main loop.
#include "gpk_internal.h"
#include <type_traits>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define BK 16
#define LDS_LD (BK + 2)

unsigned* g_cu_guest = nullptr;
hipStream_t g_chain_stream = nullptr;
unsigned long long* g_gemm_trace = nullptr;  // make DBG=1, GPK_GEMM_DBG=32: eight words per workgroup (entry, first MFMA, last MFMA, exit in 10 ns ticks; CU slot; set-up done, loads issued, loads landed)
int g_gemm_polite = 1;  // option gemm_polite: 0 = the GEMM workgroups do not look at the guest table
// this wave's slot in the guest table
static __device__ __forceinline__ unsigned cu_guest_slot() {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID: cu [11:8], sh [12], se [15:13]
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // XCC_ID [3:0]
  return ((xcc & 7u) << 8) | ((hw >> 8) & 0xffu);
}

// One block tile of the 4-wave kernel (block `bid` of the launch described by g); lds: 2 * 2 * BT * LDS_LD doubles.
template <int WT, int NS>  // WT: MFMA tiles per wave per dimension (2 -> 64x64 block); NS: register stage sets = slabs in flight
static __device__ __forceinline__ void gemm_nt_body(const GemmNT& g, const int bid, double* lds) {
  constexpr int BT = 2 * WT * 16;       // block tile edge
  constexpr int PASSES = BT / 32;       // 32 rows of 16 doubles per pass with 256 threads
  constexpr int LD = LDS_LD;
  double* As = lds;                     // [2][BT][LD]
  double* Bs = lds + 2 * BT * LD;       // [2][BT][LD]
  // launches on the panel-chain stream: their few waves share SIMDs with the bulk update's, and every microsecond of theirs is on
  // the critical path -- issue priority over the co-resident waves (N = 8192: -2 %, N = 4096: -1.5 %, profiles/r03_chain_prio.txt)
  if (g.prio) __builtin_amdgcn_s_setprio(3);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  if (g.wait_flag) {
    // resident before the producer has finished (tile pipeline: the chain kernel raises the flag in mid-flight): poll, bounded
    if (tid == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while ((int)(__hip_atomic_load(g.wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - g.wait_val) < 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 400000ull) {  // 4 ms
          if (g.wait_info) atomicCAS(g.wait_info, 0, 1 << 30);     // GPK_INFO_TIMEOUT
          break;
        }
        __builtin_amdgcn_s_sleep(16);
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }

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
  // NS slabs in flight (register set = slab index mod NS)
  d2 ra[NS][PASSES], rb[NS][PASSES];
  auto gload = [&](int kt, auto setc) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const double* pa_ = Ag + (long)(p * 32 + lrow) * g.lda + kt * BK + lchk;
      const double* pb_ = Bg + (long)(p * 32 + lrow) * g.ldb + kt * BK + lchk;
      ra[S][p] = *reinterpret_cast<const d2*>(pa_);
      rb[S][p] = *reinterpret_cast<const d2*>(pb_);
    }
  };
  // sub mode, exactly as in the 8-wave kernel below (so that every kernel of this file rounds identically and the
  // choice of tile size never changes a bit of the factor): accumulators start as C, A is negated on its way
  // into LDS, the epilogue only stores.
  const double sgn = g.sub ? -1.0 : 1.0;
  auto lstore = [&](int buf, auto setc) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      d2 na = {sgn * ra[S][p].x, sgn * ra[S][p].y};
      *reinterpret_cast<d2*>(As + (buf * BT + p * 32 + lrow) * LD + lchk) = na;
      *reinterpret_cast<d2*>(Bs + (buf * BT + p * 32 + lrow) * LD + lchk) = rb[S][p];
    }
  };
  using set0 = std::integral_constant<int, 0>;

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
  gload(0, set0{});
  if constexpr (NS > 1) { if (nk > 1) gload(1, std::integral_constant<int, 1 % NS>{}); }
  if constexpr (NS > 2) { if (nk > 2) gload(2, std::integral_constant<int, 2 % NS>{}); }
  if constexpr (NS > 3) { if (nk > 3) gload(3, std::integral_constant<int, 3 % NS>{}); }
  if constexpr (NS > 4) {
    if (nk > 4) gload(4, std::integral_constant<int, 4 % NS>{});
    if (nk > 5) gload(5, std::integral_constant<int, 5 % NS>{});
    if (nk > 6) gload(6, std::integral_constant<int, 6 % NS>{});
    if (nk > 7) gload(7, std::integral_constant<int, 7 % NS>{});
  }
  lstore(0, set0{});
  __syncthreads();
  // guest poll (gpk_internal.h), as in the 8-wave kernel: wave 0 reads its CU's counter with the slab's loads
  const unsigned* gslot = (wid == 0 && g.guest) ? g.guest + cu_guest_slot() : nullptr;
  unsigned gv = 0;
  auto slab = [&](int kt, auto setc) {  // setc: kt mod NS = the register set of slabs kt and kt + NS
    constexpr int S = decltype(setc)::value;
    using same = std::integral_constant<int, S>;
    using next = std::integral_constant<int, (S + 1) % NS>;
    const int buf = kt & 1;
    if (kt + NS < nk) gload(kt + NS, same{});  // (set S is free: slab kt went to LDS an iteration ago)
    const double* Ab = As + (buf * BT + wm * WT * 16 + fr) * LD + fg;
    const double* Bb = Bs + (buf * BT + wn * WT * 16 + fr) * LD + fg;
    {
      // one wave per SIMD and workgroup: the fragments of k-step s+1 are read under the MFMAs of k-step s (same MFMA
      // order, same bits), so that one LDS latency per slab is exposed instead of four
      double a[2][WT], b[2][WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) a[0][i] = Ab[i * 16 * LD];
#pragma unroll
      for (int j = 0; j < WT; ++j) b[0][j] = Bb[j * 16 * LD];
#pragma unroll
      for (int q = 0; q < BK / 4; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < BK / 4) {
#pragma unroll
          for (int i = 0; i < WT; ++i) a[(q + 1) & 1][i] = Ab[i * 16 * LD + 4 * (q + 1)];
#pragma unroll
          for (int j = 0; j < WT; ++j) b[(q + 1) & 1][j] = Bb[j * 16 * LD + 4 * (q + 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
          for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q & 1][i], b[q & 1][j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1, next{});
    if (gslot) {  // wave 0.  The value looked at was requested a slab ago -- behind slab kt + NS - 1, in front of slab kt + NS: it has had a
                  // whole iteration to land, and waiting for it waits for nothing this iteration would not have needed anyway (NS >= 2).
                  // Read and looked at within one iteration (the form before) every slab cost wave 0, hence the workgroup, a full
                  // round trip of an L2-coherent load: N = 8192 -2.5 %, N = 4096 -3.6 %, N = 16384 -1.3 % for moving it alone.
      if (__builtin_amdgcn_readfirstlane((int)gv) != 0) {  // a guest is on this CU: the other waves stop at the barrier
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        do {
          __builtin_amdgcn_s_sleep(32);
          gv = __hip_atomic_load(gslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (__builtin_amdgcn_readfirstlane((int)gv) != 0 && __builtin_amdgcn_s_memrealtime() - t0 < GPK_GUEST_LIMIT_TICKS);
      }
      gv = __hip_atomic_load(gslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += NS) {
    slab(kt, set0{});
    if constexpr (NS > 1) { if (kt + 1 < nk) slab(kt + 1, std::integral_constant<int, 1 % NS>{}); }
    if constexpr (NS > 2) { if (kt + 2 < nk) slab(kt + 2, std::integral_constant<int, 2 % NS>{}); }
    if constexpr (NS > 3) { if (kt + 3 < nk) slab(kt + 3, std::integral_constant<int, 3 % NS>{}); }
    if constexpr (NS > 4) {
      if (kt + 4 < nk) slab(kt + 4, std::integral_constant<int, 4 % NS>{});
      if (kt + 5 < nk) slab(kt + 5, std::integral_constant<int, 5 % NS>{});
      if (kt + 6 < nk) slab(kt + 6, std::integral_constant<int, 6 % NS>{});
      if (kt + 7 < nk) slab(kt + 7, std::integral_constant<int, 7 % NS>{});
    }
  }

  // ---- epilogue: C/D layout col = lane&15, row = (lane>>4) + 4*reg
#pragma unroll
  for (int i = 0; i < WT; ++i) {
#pragma unroll
    for (int j = 0; j < WT; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* pc_ = Cg + (long)(i * 16 + fg + 4 * r) * g.ldc + j * 16 + fr;
        *pc_ = acc[i][j][r];
      }
    }
  }
}


template <int WT, int NS>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNT g) {
  __shared__ double lds[2 * 2 * (2 * WT * 16) * LDS_LD];
  gemm_nt_body<WT, NS>(g, (int)blockIdx.x, lds);
}

// ---------------------------------------------------------------------------------------------
// ONE 128x128x128 product (the two one-tile GEMMs of a tile-pipeline step, which sit on the panel chain): 16 workgroups of
// four waves, every wave owns one 16x16 output tile and reads its 32 + 32 operand values per lane straight from L2 -- no
// LDS, no slab loop, no barrier: one load latency, then 32 MFMAs.  Same MFMA order over k, accumulators started from C,
// A negated by an exact multiplication as in the 64x64-tile kernel: the same bits.  lower: only the 16x16 tiles that
// touch the lower triangle (the leaf reads nothing above the diagonal).
__global__ __launch_bounds__(256) void gemm_tile128_kernel(GemmNT g) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int ti = (int)(blockIdx.x >> 2) * 2 + (wv >> 1), tj = (int)(blockIdx.x & 3) * 2 + (wv & 1);
  if (g.lower && tj > ti) return;
  if (g.prio) __builtin_amdgcn_s_setprio(3);
  const double* Ap = g.A + (long)(ti * 16 + fr) * g.lda + fg;
  const double* Bp = g.B + (long)(tj * 16 + fr) * g.ldb + fg;
  double* Cp = g.C + (long)(ti * 16 + fg) * g.ldc + tj * 16 + fr;
  double a[32], b[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) a[q] = Ap[4 * q];
#pragma unroll
  for (int q = 0; q < 32; ++q) b[q] = Bp[4 * q];
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  if (g.sub) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = Cp[(long)(4 * r) * g.ldc];
  }
  const double sgn = g.sub ? -1.0 : 1.0;
#pragma unroll
  for (int q = 0; q < 32; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sgn * a[q], b[q], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) Cp[(long)(4 * r) * g.ldc] = acc[r];
}

int launch_gemm_tile128(const GemmNT& g, hipStream_t s) {
  if (g.m != 128 || g.n != 128 || g.k != 128 || g.cyc_P != 0 || g.k1 != 0)
    return gpk_set_err(-1, "gemm_tile128: one 128 x 128 x 128 product only (m=%d n=%d k=%d)", g.m, g.n, g.k);
  GemmNT h = g;
  h.prio = (s == g_chain_stream) ? 1 : 0;
  hipLaunchKernelGGL(gemm_tile128_kernel, dim3(16), dim3(256), 0, s, h);
  GPK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// 8-wave 128x128-tile kernel: wave tile 64x32 (4x2 MFMA tiles, 64 accumulator VGPRs), <=104 VGPRs per wave, two
// workgroups per CU = FOUR waves per SIMD in two independent barrier domains.  The matrix pipe itself is saturated by ONE
// wave per SIMD (gpk_bench_mfma_pinned: 98-99 % of 2048 flop / 64 cycles / SIMD at W = 1, 2, 4, 8); what the second
// workgroup buys is cover for the first one's barrier, C tile and turnover (one workgroup per CU: 62 TFLOP/s, two: 66-69).
// sub mode: the accumulators START as C and A enters the MFMA negated (the instruction's neg bit), so the MFMA chain
// itself yields C - A*B^T and the epilogue is store-only.
// K-slabs are staged by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), fragments of the next
// k-step are read under the MFMAs of the current one.  Every kernel of this file accumulates over k in the same order and
// therefore rounds identically (tests: the 64x64-tile kernel is the independent cross-check of this one, bit for bit).
// Earlier forms of the slab loop (plain, register-staged + software-pipelined) and what each step bought: DESIGN.md section 6.
//
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

// SEG: the K range is the concatenation of the operand pairs of a K-segment table (gpk_internal.h: GemmSegs), each with its
// own leading dimensions -- the slab DMA walks the table with a cursor of scalar registers (base pointers, slabs left) and
// two per-lane offsets that are recomputed at a segment boundary only; the MFMA loop reads LDS and never sees the
// segments, so the result is bit for bit that of one launch per segment in table order.  NEG = (sub != 0).
// DBG (make DBG=1, timing only, wrong results): 1 no DMA in the loop, 2 no barriers, 4 no C load, 128 no C store, 8 neither, 16 every
// tile reads tile 0's first slab, 64 every workgroup uses one of 8 C tiles, 256 first-round workgroups start spread over a tile time;
// 32 (right results) per-workgroup time stamps into g.trace, +512 shader cycles of the main loop in place of the set-up stamp.  (A raised wave priority for the prologue and the store tail -- s_setprio 3 -- shortens the prologue by 2 us and lengthens the neighbour's slab loop by as much: profiles/r03_gemm_tile_trace.txt.)
struct NoSegs {};
template <bool SEG> struct SegArg { typedef NoSegs type; };
template <> struct SegArg<true> { typedef GemmSegs type; };
// one 128x128 tile (tile number `bid` of the launch) by the 512 threads of a workgroup; lds: 2 * 2 * 128 * BK doubles
template <bool SEG, bool NEG, int DBG>
static __device__ __forceinline__ void gemm_nt8_tile(const GemmNT& g, const typename SegArg<SEG>::type& segs, double* lds, const unsigned bid) {
  constexpr int BT = 128;
  unsigned long long st_entry = 0, st_first = 0, st_last = 0, st_a = 0, st_b = 0, st_c = 0;
  if (DBG & 32) st_entry = __builtin_amdgcn_s_memrealtime();
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
    const int xcd = bid & 7, seq = bid >> 3;
    const int sq = (seq >> 6) * 8 + xcd, slot = seq & 63;
    const int SN = (ntn + 7) >> 3;
    int I, J;
    if (g.lower) {  // lower trapezoid of squares: SN x SN triangle, then full rows of SN squares
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
    int id = bid;
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
    ti = bid / ntn;
    tj = bid % ntn;
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
  const int fr = lane & 15, fg = lane >> 4;
  double* Cw = (DBG & 64) ? g.C + ((long)(bid & 7) * BT + wm * 64) * g.ldc + wn * 32  // DBG 64: every workgroup uses one of 8 C tiles (cache hits)
                          : g.C + ((long)ti * BT + wm * 64) * g.ldc + (long)tj * BT + wn * 32;  // wave-uniform
  const unsigned c_off = (unsigned)fg * (unsigned)g.ldc + (unsigned)fr;

  d4 acc[4][2];
  auto init_acc = [&]() {
    if (g.sub && !(DBG & 8) && !(DBG & 4)) {
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

  const int nk = g.k / BK;  // (SEG: the host sets g.k to the sum of the segments)
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
  // DMA cursor: slabs are requested strictly in order, so the source is a pair of scalar pointers that advance by one slab
  // per request; with a segment table they are reloaded (with the leading dimensions and the per-lane offsets that depend
  // on them) when a segment runs out -- a scalar branch taken once per segment, not per slab.
  const double *cAp = Ag, *cBp = Bg;
  unsigned voA = (unsigned)(((long)rr * g.lda + gch) * 8);  // < 4 GiB: 128 rows of one operand (host check)
  unsigned voB = (unsigned)(((long)rr * g.ldb + gch) * 8);
  long a64 = 64 * g.lda, b64 = 64 * g.ldb;
  int seg_i = 0, seg_left = nk;
  auto seg_open = [&](int i) {
    if constexpr (SEG) {
      const GemmSeg& sg = segs.s[i];
      cAp = sg.A + (long)ti * BT * sg.lda;
      cBp = sg.B + brow0 * sg.ldb;
      voA = (unsigned)(((long)rr * sg.lda + gch) * 8);
      voB = (unsigned)(((long)rr * sg.ldb + gch) * 8);
      a64 = 64 * (long)sg.lda;
      b64 = 64 * (long)sg.ldb;
      seg_left = sg.k / BK;
    }
  };
  if constexpr (SEG) seg_open(0);
  auto dma = [&](int buf) {
    const unsigned dst = lds0 + (unsigned)((buf * STG + wid * 8 * BK) * 8);
    glds16(cAp, voA, dst);
    glds16(cAp + a64, voA, dst + 64 * BK * 8);
    glds16(cBp, voB, dst + OPD * 8);
    glds16(cBp + b64, voB, dst + OPD * 8 + 64 * BK * 8);
    if (!(DBG & 16)) {  // DBG 16: every request reads the first slab
      cAp += BK;
      cBp += BK;
    }
    if constexpr (SEG) {
      if (--seg_left == 0 && ++seg_i < segs.n) seg_open(seg_i);
    }
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
  // the fragment set `slot` is in registers once at most six younger LDS reads are outstanding
  auto landed6 = [&](int slot) {
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(pa[slot][0]), "+v"(pa[slot][1]), "+v"(pa[slot][2]), "+v"(pa[slot][3]), "+v"(pb[slot][0]), "+v"(pb[slot][1]));
  };
  auto landed0 = [&](int slot) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pa[slot][0]), "+v"(pa[slot][1]), "+v"(pa[slot][2]), "+v"(pa[slot][3]), "+v"(pb[slot][0]), "+v"(pb[slot][1]));
  };
  constexpr int NEGA = NEG ? 1 : 0;
  auto mm = [&](int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[slot][i], pb[slot][j], acc[i][j], 0, 0, NEGA);
  };
  // guest poll (gpk_internal.h): wave 0 only; the value read during one slab is looked at after the next barrier
  const unsigned* gslot = (wid == 0 && g.guest) ? g.guest + cu_guest_slot() : nullptr;
  unsigned gv = 0;
  if ((DBG & 256) && bid < 512) {  // DBG 256: first-round workgroups start spread over one tile time (is the C hand-over a chip-wide convoy?)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long dl = (unsigned long long)(((bid * 2654435761u) >> 29) & 7u) * (unsigned long long)g.k * 3ull;  // x/8 of 0.234 us * k, in 10 ns ticks
    while (__builtin_amdgcn_s_memrealtime() - t0 < dl) __builtin_amdgcn_s_sleep(8);
  }
  if (DBG & 32) st_a = __builtin_amdgcn_s_memrealtime();
  dma(0);
  if (nk > 1) dma(1);
  init_acc();  // (the first two slabs are requested BEFORE the C tile, which queues behind them in the load path)
  if (DBG & 32) st_b = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (DBG & 32) st_c = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  if (DBG & 32) st_first = __builtin_amdgcn_s_memrealtime();
  unsigned long long cyc0 = 0;
  if (DBG & 512) cyc0 = __builtin_amdgcn_s_memtime();
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
    if (!(DBG & 1) && kt + 2 < nk) dma(buf);
    if (gslot && (kt & g.poll_mask) == g.poll_mask) {  // scalar branch: wave 0
      if (__builtin_amdgcn_readfirstlane((int)gv) != 0) {
        // a guest (the leaf) is on this CU: stay out of its way.  The other seven waves stop at the next barrier.
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        do {
          __builtin_amdgcn_s_sleep(32);
          gv = __hip_atomic_load(gslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (__builtin_amdgcn_readfirstlane((int)gv) != 0 && __builtin_amdgcn_s_memrealtime() - t0 < GPK_GUEST_LIMIT_TICKS);
      }
      gv = __hip_atomic_load(gslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // lands under this slab's MFMAs
    }
  }
  if (DBG & 32) st_last = __builtin_amdgcn_s_memrealtime();
  if (DBG & 512) st_a = __builtin_amdgcn_s_memtime() - cyc0;  // shader cycles of the main loop instead of the set-up stamp
  if (DBG & (8 | 128)) {
    double t = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) t += acc[i][j][r];
    if (t == 12345.678) Cw[c_off] = t;
    if ((DBG & 32) && g.trace && tid == 0) {
      unsigned long long* tr = g.trace + 8 * (size_t)bid;
      tr[0] = st_entry; tr[1] = st_first; tr[2] = st_last; tr[3] = __builtin_amdgcn_s_memrealtime(); tr[4] = cu_guest_slot();
      tr[5] = st_a; tr[6] = st_b; tr[7] = st_c;
    }
    return;
  }
  // Write-through (sc1) stores: the C tiles are not re-read before the next panel step, and lines left dirty in
  // the XCD L2s make EVERY kernel boundary on the chip pay for their write-back (the end-of-kernel release of the
  // latency-critical leaf went from 60 us alone to 155-173 us beside this kernel).
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        __hip_atomic_store((Cw + (long)(i * 16 + 4 * r) * g.ldc + j * 16) + c_off, acc[i][j][r], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  if ((DBG & 32) && g.trace && tid == 0) {
    unsigned long long* t = g.trace + 8 * (size_t)bid;
    t[0] = st_entry; t[1] = st_first; t[2] = st_last; t[3] = __builtin_amdgcn_s_memrealtime(); t[4] = cu_guest_slot();
    t[5] = st_a; t[6] = st_b; t[7] = st_c;
  }
}


template <bool SEG, bool NEG, int DBG = 0>
__global__ __launch_bounds__(512, 4) void gemm_nt8_kernel(GemmNT g, typename SegArg<SEG>::type segs) {
  // the unpadded LDS image, exactly 64 KiB: two workgroups per CU leave 32 KiB, and ONE finished workgroup frees a 96 KiB
  // slot, which is what the slot-sized leaf (k_potf2.hip) is built to fit
  __shared__ double lds[2 * 2 * 128 * BK];
  gemm_nt8_tile<SEG, NEG, DBG>(g, segs, lds, blockIdx.x);
}

// Persistent form for the deferred corner updates (K in the thousands, a tile takes half a millisecond): a grid SMALLER than
// the chip's 512 workgroup slots -- by default one workgroup per CU -- whose workgroups take tiles from a counter until
// none is left.  The slots it does not take stay free for the whole launch: the panel chain's kernels (and the short
// K = 256 updates of the update stream) find one at once instead of waiting for a half-millisecond tile to retire, which is
// what made the chain 30 % slower beside a plain grid of such tiles.  Same tile code, same bits.
template <bool NEG>
__global__ __launch_bounds__(512, 4) void gemm_nt8_persistent_kernel(GemmNT g, GemmSegs segs, unsigned* counter, unsigned ntiles) {
  __shared__ double lds[2 * 2 * 128 * BK];
  __shared__ unsigned next_tile;
  for (;;) {
    if (threadIdx.x == 0) next_tile = atomicAdd(counter, 1u);
    __syncthreads();  // (also: every wave has left the previous tile's last LDS reads behind)
    const unsigned t = next_tile;
    __syncthreads();
    if (t >= ntiles) return;
    gemm_nt8_tile<true, NEG, 0>(g, segs, lds, t);
  }
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
  if (g.wait_flag && tile128) return gpk_set_err(-1, "gemm_nt: only the 64x64-tile kernel polls a flag");
  if (g.k1 > 0) {  // two separately stored operands: a table of two segments
    if (!tile128 || g.cyc_P != 0) return gpk_set_err(-1, "gemm_nt: a second K segment needs the 8-wave single-GPU kernel");
    if (g.k1 >= g.k || !g.A2 || !g.B2) return gpk_set_err(-1, "gemm_nt: bad second K segment (k1=%d k=%d)", g.k1, g.k);
    GemmSegs sg{};
    sg.n = 2;
    sg.s[0] = GemmSeg{g.A, g.B, (int)g.lda, (int)g.ldb, g.k1, 0};
    sg.s[1] = GemmSeg{g.A2, g.B2, (int)g.lda2, (int)g.ldb2, g.k - g.k1, 0};
    return launch_gemm_nt_segs(g, sg, s);
  }
  if (!g.guest && g_cu_guest && g_gemm_polite) {
    GemmNT h = g;
    h.guest = g_cu_guest;
    h.poll_mask = 3;  // every fourth slab (~5 us): the poll costs the 8-wave kernel 1 % at K = 512 when every slab asks (profiles/r04_skinny_gemm_poll.txt)
    return launch_gemm_nt(h, tile128, s);
  }
  const int ntm = g.m / BT, ntn = g.n / BT;
  long nblk;
  if (g.lower && g.cyc_P == 0) {
    if (g.m < g.n) return gpk_set_err(-1, "gemm_nt: lower needs m >= n (m=%d n=%d)", g.m, g.n);
    nblk = (long)ntn * (ntn + 1) / 2 + (long)(ntm - ntn) * ntn;
  } else {
    nblk = (long)ntm * ntn;
  }
  if (!tile128) {
    GemmNT h = g;
    h.prio = (s == g_chain_stream) ? 1 : 0;
    // two slabs in flight (one, three, four, eight measured the same once the guest poll was out of the slab's dependent path:
    // profiles/r04_skinny_gemm_poll.txt)
    hipLaunchKernelGGL((gemm_nt_kernel<2, 2>), dim3((unsigned)nblk), dim3(256), 0, s, h);
    GPK_HIP(hipGetLastError());
    return 0;
  }
  if (g.xcd_sq && g.cyc_P == 0) {  // whole 8x8 squares, 8 squares (one per XCD) per round
    const long SM = (ntm + 7) / 8, SN = (ntn + 7) / 8;
    const long nsq = g.lower ? SN * (SN + 1) / 2 + (SM - SN) * SN : SM * SN;
    nblk = ((nsq + 7) / 8) * 8 * 64;
  }
  const dim3 grid((unsigned)nblk), block(512);
#ifdef GPK_GEMM_DBG_VARIANTS
  // timing-only builds of the kernel (make DBG=1; scripts/gemm_dbg.py): GPK_GEMM_DBG bits 1 no DMA in the loop, 2 no
  // barriers, 8 no C load / store, 16 every tile reads tile 0's first slab; GPK_GEMM_DYNLDS=<bytes> of extra LDS per
  // workgroup (40960: one workgroup per CU).  Wrong results.
  static const int dbg = getenv("GPK_GEMM_DBG") ? atoi(getenv("GPK_GEMM_DBG")) : 0;
  static const int dyn = getenv("GPK_GEMM_DYNLDS") ? atoi(getenv("GPK_GEMM_DYNLDS")) : 0;
  if ((dbg || dyn) && g.k1 == 0 && g.sub) {
#define GPK_DBG_LAUNCH(D)                                                                                                 \
  do {                                                                                                                    \
    if (dyn > 0) (void)hipFuncSetAttribute((const void*)gemm_nt8_kernel<false, true, D>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn); \
    hipLaunchKernelGGL((gemm_nt8_kernel<false, true, D>), grid, block, dyn, s, g, NoSegs{});                              \
  } while (0)
    switch (dbg) {
      case 1: GPK_DBG_LAUNCH(1); break;
      case 2: GPK_DBG_LAUNCH(2); break;
      case 8: GPK_DBG_LAUNCH(8); break;
      case 9: GPK_DBG_LAUNCH(9); break;
      case 16: GPK_DBG_LAUNCH(16); break;
      case 64: GPK_DBG_LAUNCH(64); break;
      case 256: GPK_DBG_LAUNCH(256); break;
      case 4: GPK_DBG_LAUNCH(4); break;
      case 128: GPK_DBG_LAUNCH(128); break;
      case 32: { GemmNT gt = g; gt.trace = g_gemm_trace; hipLaunchKernelGGL((gemm_nt8_kernel<false, true, 32>), grid, block, dyn, s, gt, NoSegs{}); } break;
      case 544: { GemmNT gt = g; gt.trace = g_gemm_trace; hipLaunchKernelGGL((gemm_nt8_kernel<false, true, 544>), grid, block, dyn, s, gt, NoSegs{}); } break;
      case 552: { GemmNT gt = g; gt.trace = g_gemm_trace; hipLaunchKernelGGL((gemm_nt8_kernel<false, true, 552>), grid, block, dyn, s, gt, NoSegs{}); } break;
      default: GPK_DBG_LAUNCH(0); break;
    }
#undef GPK_DBG_LAUNCH
    GPK_HIP(hipGetLastError());
    return 0;
  }
#endif
  if (g.sub) hipLaunchKernelGGL((gemm_nt8_kernel<false, true>), grid, block, 0, s, g, NoSegs{});
  else hipLaunchKernelGGL((gemm_nt8_kernel<false, false>), grid, block, 0, s, g, NoSegs{});
  GPK_HIP(hipGetLastError());
  return 0;
}

int launch_gemm_nt_segs(const GemmNT& g0, const GemmSegs& segs, hipStream_t s, int persist_grid, unsigned* counter) {
  if (g0.m <= 0 || g0.n <= 0) return 0;
  if (g0.m % 128 || g0.n % 128) return gpk_set_err(-1, "gemm_nt_segs: bad shape m=%d n=%d", g0.m, g0.n);
  if (g0.cyc_P != 0) return gpk_set_err(-1, "gemm_nt_segs: single-GPU column map only");
  if (segs.n < 1 || segs.n > GPK_MAXSEG) return gpk_set_err(-1, "gemm_nt_segs: %d segments (1..%d)", segs.n, GPK_MAXSEG);
  long ktot = 0;
  for (int i = 0; i < segs.n; ++i) {
    const GemmSeg& q = segs.s[i];
    if (q.k <= 0 || q.k % BK || !q.A || !q.B || (q.lda & 1) || (q.ldb & 1) || q.lda < q.k || q.ldb < q.k || q.lda >= (1 << 22) || q.ldb >= (1 << 22) ||
        ((uintptr_t)q.A & 15) || ((uintptr_t)q.B & 15))
      return gpk_set_err(-1, "gemm_nt_segs: bad segment %d (k=%d lda=%d ldb=%d)", i, q.k, q.lda, q.ldb);
    ktot += q.k;
  }
  if (ktot >= (1l << 30)) return gpk_set_err(-1, "gemm_nt_segs: K too large");
  GemmNT g = g0;
  g.k = (int)ktot;
  g.k1 = 0;
  g.A = segs.s[0].A; g.lda = segs.s[0].lda;  // (unused by the kernel in table mode; kept consistent for the tile decode)
  g.B = segs.s[0].B; g.ldb = segs.s[0].ldb;
  g.A2 = g.B2 = nullptr;
  if (!g.guest && g_cu_guest && g_gemm_polite) {
    g.guest = g_cu_guest;
    g.poll_mask = 3;
  }
  const int ntm = g.m / 128, ntn = g.n / 128;
  long nblk;
  if (g.lower) {
    if (g.m < g.n) return gpk_set_err(-1, "gemm_nt_segs: lower needs m >= n (m=%d n=%d)", g.m, g.n);
    nblk = (long)ntn * (ntn + 1) / 2 + (long)(ntm - ntn) * ntn;
  } else {
    nblk = (long)ntm * ntn;
  }
  if (g.xcd_sq) {
    const long SM = (ntm + 7) / 8, SN = (ntn + 7) / 8;
    const long nsq = g.lower ? SN * (SN + 1) / 2 + (SM - SN) * SN : SM * SN;
    nblk = ((nsq + 7) / 8) * 8 * 64;
  }
  if (persist_grid > 0 && counter && !g.xcd_sq && nblk > persist_grid) {
    // persistent: persist_grid workgroups share the nblk tiles through `counter` (zeroed here, on the same stream)
    GPK_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned), s));
    const dim3 pgrid((unsigned)persist_grid), block(512);
    if (g.sub) hipLaunchKernelGGL((gemm_nt8_persistent_kernel<true>), pgrid, block, 0, s, g, segs, counter, (unsigned)nblk);
    else hipLaunchKernelGGL((gemm_nt8_persistent_kernel<false>), pgrid, block, 0, s, g, segs, counter, (unsigned)nblk);
    GPK_HIP(hipGetLastError());
    return 0;
  }
  const dim3 grid((unsigned)nblk), block(512);
  if (g.sub) hipLaunchKernelGGL((gemm_nt8_kernel<true, true>), grid, block, 0, s, g, segs);
  else hipLaunchKernelGGL((gemm_nt8_kernel<true, false>), grid, block, 0, s, g, segs);
  GPK_HIP(hipGetLastError());
  return 0;
}
