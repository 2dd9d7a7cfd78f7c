// Persistent, software-pipelined fp64 MFMA kernel for the short-K rank-NB updates
//     C[m x n] -= A[m x k] * B[n x k]^T        (k = panel width, 256..512)
// i.e. the Cholesky trailing update (A = B = panel, lower tiles only) and the forward-
// substitution update of the prediction solve.  With K this short a 128x128 tile is only
// ~65k MFMA cycles, so the plain tile-per-workgroup kernel (k_gemm.hip) spends a third of
// its time in the prologue and in the C read-modify-write epilogue (per-CU HBM rate
// ~10 B/clk).  Here every fixed cost is overlapped with the matrix pipe:
//
//   * one workgroup of EIGHT waves per CU walks a strided list of tiles (persistent).  Measured on
//     MI355X: a single wave per SIMD issues v_mfma_f64_16x16x4_f64 at only ~30% of peak, two
//     waves per SIMD reach ~80%, so the design is pinned to 2 waves/SIMD = 256 VGPRs per wave;
//   * wave tile 64x32 (4x2 MFMA tiles), TWO accumulator sets (2 x 64 VGPRs): while tile t
//     accumulates in one set, the other set is first stored (results of tile t-1) and then
//     refilled with the C values of tile t+1, a few memory instructions per K-slab, all
//     hidden behind MFMAs.  acc starts as C and A is negated on its way into LDS, so the
//     MFMA chain itself produces C - A*B^T: the epilogue is store-only;
//   * the A/B K-slabs (16 deep) stream global -> registers -> LDS through THREE LDS buffers
//     with one raw s_barrier per slab, placed before the last quarter of the slab so that the
//     first fragments of the next slab are prefetched behind the current slab's MFMAs;
//     the slab pipeline runs straight across tile boundaries;
//   * fragments are double-buffered in registers (ds_read of step k+1 behind MFMAs of step k).
//
// Layouts as in k_gemm.hip: A/B fragment lane l = M[r0 + (l&15)][k0 + (l>>4)];
// C/D (f64): col = lane&15, row = (lane>>4) + 4*reg; LDS row stride 18 doubles (conflict-free).
#include "gpk_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define BK 16
#define LLD 18
#define TILE 128
#define SLAB (2 * TILE * LLD)  // doubles per LDS buffer: A rows then B rows

struct TileRef {
  const double* A;  // first row of the A tile
  const double* B;  // first row of the B tile
  double* C;        // wave-local C origin is derived from this (tile origin)
  int valid;
};

__device__ __forceinline__ TileRef tile_ref(const GemmNT& g, int id, int ntiles) {
  TileRef t;
  t.valid = id < ntiles;
  if (!t.valid) id = ntiles - 1;
  int ti, tj;
  if (g.lower) {
    int r = (int)((sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
    while (r * (r + 1) / 2 > id) --r;
    while ((r + 1) * (r + 2) / 2 <= id) ++r;
    ti = r;
    tj = id - r * (r + 1) / 2;
  } else {
    const int ntn = g.n / TILE;
    ti = id / ntn;
    tj = id - ti * ntn;
  }
  ti = __builtin_amdgcn_readfirstlane(ti);  // wave-uniform by construction: keep the pointers in SGPRs
  tj = __builtin_amdgcn_readfirstlane(tj);
  t.A = g.A + (long)ti * TILE * g.lda;
  t.B = g.B + (long)tj * TILE * g.ldb;
  t.C = g.C + (long)ti * TILE * g.ldc + (long)tj * TILE;
  return t;
}

struct PipeState {
  // slab loader (runs two slabs ahead of the MFMAs, across tile boundaries)
  const double* la;  // A source of the next slab to load (wave-uniform: tile origin + slab offset)
  const double* lb;
  int l_kt;          // slab index inside its tile
  int l_live;        // loader still has slabs to fetch
  int wbuf;          // LDS buffer the next ds_write goes to
  int rbuf;          // LDS buffer the MFMAs read
};

template <int NK8>  // slabs per phase: K = 16 * 8 * NK8  (NK8 = 2 -> K = 256, 4 -> K = 512)
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(GemmNT g, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) double lds[];  // 3 * SLAB doubles
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;  // 2 x 4 waves, each 64 x 32
  const int fr = lane & 15, fg = lane >> 4;
  const int lrow = tid >> 3, lchk = (tid & 7) * 2;  // 64 rows x 8 chunks per pass
  const int G = gridDim.x;
  constexpr int NK = 8 * NK8;

  d4 acc0[4][2], acc1[4][2];
  d2 ra[2][2], rb[2][2];  // [register set][pass]: two slabs in flight from global memory
  double fa[2][4], fb[2][2];

  int cur_id = blockIdx.x;
  TileRef cur = tile_ref(g, cur_id, ntiles);
  if (!cur.valid) return;

  TileRef next = tile_ref(g, cur_id + G, ntiles);
  PipeState st;
  st.l_kt = 0;
  st.l_live = 1;
  st.la = cur.A;
  st.lb = cur.B;
  st.wbuf = 0;
  st.rbuf = 0;

  // per-lane element offsets (32-bit, added to wave-uniform bases: saddr + voffset addressing)
  const unsigned la_off = (unsigned)lrow * (unsigned)g.lda + (unsigned)lchk;
  const unsigned lb_off = (unsigned)lrow * (unsigned)g.ldb + (unsigned)lchk;
  const unsigned c_off = (unsigned)fg * (unsigned)g.ldc + (unsigned)fr;
  auto gload = [&](d2 (&xa)[2], d2 (&xb)[2]) {  // fetch the loader's current slab and advance the loader
    if (st.l_live && !(g.dbg & 2)) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        xa[p] = *reinterpret_cast<const d2*>(st.la + (long)(p * 64) * g.lda + la_off);
        xb[p] = *reinterpret_cast<const d2*>(st.lb + (long)(p * 64) * g.ldb + lb_off);
      }
      st.la += BK;
      st.lb += BK;
      if (++st.l_kt == NK) {  // the loader is at most two slabs ahead: the tile it enters is `next`
        st.l_kt = 0;
        st.l_live = next.valid;
        st.la = next.A;
        st.lb = next.B;
      }
    }
  };
  auto lstore = [&](d2 (&xa)[2], d2 (&xb)[2]) {  // regs -> LDS buffer wbuf (A negated: the MFMA chain then yields C - A*B^T)
    double* dA = lds + st.wbuf * SLAB + lrow * LLD + lchk;
    double* dB = dA + TILE * LLD;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      d2 na = {-xa[p].x, -xa[p].y};
      *reinterpret_cast<d2*>(dA + p * 64 * LLD) = na;
      *reinterpret_cast<d2*>(dB + p * 64 * LLD) = xb[p];
    }
    st.wbuf = (st.wbuf == 2) ? 0 : st.wbuf + 1;
  };
  auto frag_read = [&](int buf, int kk, double (&a)[4], double (&b)[2]) {
    const double* Ab = lds + buf * SLAB + (wm * 64 + fr) * LLD + fg + kk;
    const double* Bb = lds + buf * SLAB + TILE * LLD + (wn * 32 + fr) * LLD + fg + kk;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = Ab[i * 16 * LLD];
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = Bb[j * 16 * LLD];
  };
  auto lds_barrier = [&]() {
    if (g.dbg & 4) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  };
  // C element (i, j, r) of this lane for a tile with origin C0
  auto cptr = [&](double* C0, int i, int j, int r) -> double* {
    double* row = C0 + (long)(wm * 64 + i * 16 + 4 * r) * g.ldc + wn * 32 + j * 16;  // wave-uniform
    return row + c_off;
  };

  // ---------------- pipeline prologue: C(cur) -> acc0, slabs 0 and 1 staged
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc0[i][j][r] = *cptr(cur.C, i, j, r);
  gload(ra[0], rb[0]);   // slab 0
  lstore(ra[0], rb[0]);  // -> buf 0
  gload(ra[1], rb[1]);   // slab 1 (written to LDS in iteration 0)
  gload(ra[0], rb[0]);   // slab 2 (written to LDS in iteration 1)
  lds_barrier();
  frag_read(0, 0, fa[0], fb[0]);

  // one K-slab: MFMAs of slab `it` from rbuf; LDS write of slab it+1; global fetch of slab it+2
  // `xa/xb` is the register set holding slab it+1 (fetched TWO iterations ago: one slab of MFMAs does
  // not cover the loaded-chip L2/MALL latency for a lone workgroup per CU); it is refilled with slab it+3.
  auto slab = [&](d4 (&acc)[4][2], d2 (&xa)[2], d2 (&xb)[2]) {
    if (!(g.dbg & 8)) lstore(xa, xb);
    gload(xa, xb);
    const int rb_ = st.rbuf;
    const int nb_ = (rb_ == 2) ? 0 : rb_ + 1;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int cb = k4 & 1, nbf = cb ^ 1;
      if (k4 == 3) {
        lds_barrier();                          // slab it+1 visible to every wave
        frag_read(nb_, 0, fa[nbf], fb[nbf]);    // first fragments of the next slab
      } else {
        frag_read(rb_, (k4 + 1) * 4, fa[nbf], fb[nbf]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cb][i], fb[cb][j], acc[i][j], 0, 0, 0);
    }
    st.rbuf = nb_;
  };

  // one tile = 8 phases x NK8 slabs.  Phase p < 4 first stores rows-group p of the PREVIOUS tile's
  // result (held in `oth`); phase p >= 4 first loads rows-group p-4 of the NEXT tile's C into `oth`.
  auto run_tile = [&](d4 (&acc)[4][2], d4 (&oth)[4][2], const TileRef& prev, const TileRef& next) {
#pragma unroll
    for (int ph = 0; ph < 8; ++ph) {
      if (!(g.dbg & 1)) {
        if (ph < 4) {
          if (prev.valid) {
            const int i = ph;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) *cptr(prev.C, i, j, r) = oth[i][j][r];
          }
        } else {
          if (next.valid) {
            const int i = ph - 4;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) oth[i][j][r] = *cptr(next.C, i, j, r);
          }
        }
      }
#pragma nounroll
      for (int q = 0; q < NK8 / 2; ++q) {
        slab(acc, ra[1], rb[1]);
        slab(acc, ra[0], rb[0]);
      }
    }
  };

  TileRef prev;
  prev.valid = 0;
  prev.A = prev.B = nullptr;
  prev.C = nullptr;
  while (true) {
    run_tile(acc0, acc1, prev, next);
    prev = cur;
    cur = next;
    cur_id += G;
    if (!cur.valid) {  // drain: results of the last tile are in acc0
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) *cptr(prev.C, i, j, r) = acc0[i][j][r];
      break;
    }
    next = tile_ref(g, cur_id + G, ntiles);
    run_tile(acc1, acc0, prev, next);
    prev = cur;
    cur = next;
    cur_id += G;
    if (cur.valid) next = tile_ref(g, cur_id + G, ntiles);
    if (!cur.valid) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) *cptr(prev.C, i, j, r) = acc1[i][j][r];
      break;
    }
  }
}

int launch_gemm_nt_pp(const GemmNT& g, int ncu, hipStream_t s) {
  if (g.m <= 0 || g.n <= 0) return 0;
  if (!g.sub || g.cyc_P != 0) return gpk_set_err(-1, "gemm_nt_pp: only the C -= A*B^T, single-GPU form");
  if (g.m % TILE || g.n % TILE || (g.k != 256 && g.k != 512))
    return gpk_set_err(-1, "gemm_nt_pp: bad shape m=%d n=%d k=%d", g.m, g.n, g.k);
  if ((g.lda & 1) || (g.ldb & 1) || ((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15))
    return gpk_set_err(-1, "gemm_nt_pp: operands must be 16-byte aligned with even leading dimensions");
  const long ntm = g.m / TILE, ntn = g.n / TILE;
  long ntiles;
  if (g.lower) {
    if (g.m != g.n) return gpk_set_err(-1, "gemm_nt_pp: lower needs a square C");
    ntiles = ntm * (ntm + 1) / 2;
  } else {
    ntiles = ntm * ntn;
  }
  if (ntiles > 0x3fffffff) return gpk_set_err(-1, "gemm_nt_pp: too many tiles");
  const long grid = ntiles < ncu ? ntiles : ncu;
  const size_t shmem = (size_t)3 * SLAB * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    GPK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_pp_kernel<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    GPK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_pp_kernel<4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    attr_set = true;
  }
  if (g.k == 256)
    hipLaunchKernelGGL(gemm_nt_pp_kernel<2>, dim3((unsigned)grid), dim3(512), shmem, s, g, (int)ntiles);
  else
    hipLaunchKernelGGL(gemm_nt_pp_kernel<4>, dim3((unsigned)grid), dim3(512), shmem, s, g, (int)ntiles);
  GPK_HIP(hipGetLastError());
  return 0;
}
