// Squared-exponential kernel-matrix builder (and the plain squared distance).
// Replaces squared_distance (GPr.py:4-13), SquaredExponential.compute_Kxx_matrix
// (GPr.py:99-103), compute_Kxz_matrix (GPr.py:105-110) and GPy's RBF-ARD K
// (call site GPpref.py:122).
//
// HBM-write-bound: 8 bytes per output element against (3d+~40) fp64 ops.  One
// workgroup = one 128x128 output tile.  The 128 scaled row points are staged in
// LDS once (read back as wave-wide broadcasts), each lane keeps its TWO column
// points in registers, and every wave instruction stores one full 1 KiB row
// segment (64 lanes x 16 B), so the stores are perfectly coalesced.  The
// reference's A^2+B^2-2AB form needs four N^2 temporaries; here the distance is
// accumulated in registers: as sum((a-b)^2) in general, and for d <= 8 in the reference's own dot-product form on points
// CENTRED on one of them (differences are unchanged, and the cancellation error eps * |x/ell|^2 of that form is then
// governed by the spread of the data, not by its offset from the origin).
#include "gpk_internal.h"

typedef double d2 __attribute__((ext_vector_type(2)));

// 2^t for t <= ~10 (the SE kernel: t = log2(sf2) - r^2 log2(e)/2), ~1 ulp: round to the nearest integer, degree-12
// Taylor polynomial of 2^f on |f| <= 1/2 (truncation 1.7e-16 relative), exponent added with v_ldexp (gradual underflow
// to 0 for very negative t comes for free).  ~18 VALU operations against ~30 of the library exp() with its range and
// special-value handling: with d = 8 the builder is VALU-bound on exactly that difference.
__device__ __forceinline__ double exp2_neg(double t) {
  t = (t < -1100.0) ? -1100.0 : t;  // not fmax: a NaN input must stay NaN (non-finite points have to fail the factorisation)
  const double n = rint(t);
  const double f = t - n;
  double p = 2.5678435993488196e-11;   // ln2^12/12!  (degree 12: truncation 1.7e-16 relative on |f| <= 1/2)
  p = fma(p, f, 4.44553827187081e-10);
  p = fma(p, f, 7.054911620801121e-09);
  p = fma(p, f, 1.0178086009239696e-07);
  p = fma(p, f, 1.3215486790144305e-06);
  p = fma(p, f, 1.5252733804059838e-05);
  p = fma(p, f, 0.00015403530393381606);
  p = fma(p, f, 0.0013333558146428441);
  p = fma(p, f, 0.009618129107628477);
  p = fma(p, f, 0.055504108664821576);
  p = fma(p, f, 0.2402265069591007);
  p = fma(p, f, 0.6931471805599453);
  p = fma(p, f, 1.0);
  return ldexp(p, (int)n);
}

// MODE 0: sf2 exp(-r^2/2) (+ diag_add on the diagonal of a same-set build), 1: plain r^2, 2: stores only (timing).
// In mode 0 the points are pre-scaled by sqrt(log2(e)/2)/ell so that the accumulated squared distance IS the (negated)
// base-2 exponent, and log2(sf2) is folded into it: one add + exp2_neg per element instead of two multiplies + exp().
template <int D, int MODE>  // D = 0: runtime dimension (column points re-read from LDS)
__global__ __launch_bounds__(256) void se_build_kernel(const double* __restrict__ R, int nr,
                                                       const double* __restrict__ Cc, int nc, SEParams p,
                                                       int same_set, int lower_only,
                                                       double* __restrict__ out, long ld, int col_g0,
                                                       const double* __restrict__ centre) {
  constexpr int DMAX = (D == 0) ? 32 : D;
  __shared__ double rs[128 * DMAX];
  __shared__ double cs[(D == 0) ? 128 * DMAX : 1];
  __shared__ double rn[128];  // squared norms of the (scaled) row points (mode 0, D != 0)
  const int d = (D == 0) ? p.d : D;
  int tc = blockIdx.x, tr = blockIdx.y;
  if (lower_only == 2) {  // 1-D grid over the lower triangle of tiles (no workgroups that only exit)
    const int id = blockIdx.x;
    int t = (int)((sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
    while (t * (t + 1) / 2 > id) --t;
    while ((t + 1) * (t + 2) / 2 <= id) ++t;
    tr = t;
    tc = id - t * (t + 1) / 2;
  }
  // col_g0: global index of column 0 of this call (block-cyclic local column blocks); Cc/out are already offset
  if (lower_only == 1 && col_g0 + tc * 128 > tr * 128 + 127) return;
  const int r0 = tr * 128, c0 = tc * 128;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const double pre = (MODE == 0) ? 0.8493218002880191 : 1.0;  // sqrt(log2(e) / 2)
  const double lsf = (MODE == 0) ? log2(p.sf2) : 0.0;
  constexpr bool DOTFORM = (MODE == 0 && D != 0);
  double cen[DMAX];  // dot-product form: both point sets are centred on one point (exact for the differences)
#pragma unroll
  for (int k = 0; k < DMAX; ++k) cen[k] = (DOTFORM && k < d) ? centre[k] : 0.0;

  for (int e = tid; e < 128 * d; e += 256) {
    const int i = e / d, k = e - i * d;
    const int gi = r0 + i;
    rs[e] = (gi < nr) ? (R[(long)gi * d + k] - (DOTFORM ? centre[k] : 0.0)) / p.ell[k] * pre : 0.0;
    if (D == 0) {
      const int gc = c0 + i;
      cs[e] = (gc < nc) ? Cc[(long)gc * d + k] / p.ell[k] * pre : 0.0;
    }
  }
  const int ca = c0 + 2 * lane, cb = ca + 1;
  double za[DMAX], zb[DMAX];
  // Mode 0 with register-resident column points: the exponent as |a|^2 + |b|^2 - 2 a.b (the very form of the
  // reference's squared_distance, GPr.py:4-13), 10 operations per element instead of 16; the norms are accumulated
  // in the same order as the dot product, so the diagonal is exactly 0.  Absolute error of the exponent ~1e-15.
  double na = 0.0, nb = 0.0;
  if (D != 0) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      za[k] = (ca < nc) ? (Cc[(long)ca * D + k] - cen[k]) / p.ell[k] * pre : 0.0;
      zb[k] = (cb < nc) ? (Cc[(long)cb * D + k] - cen[k]) / p.ell[k] * pre : 0.0;
      na = fma(za[k], za[k], na);
      nb = fma(zb[k], zb[k], nb);
    }
  }
  __syncthreads();
  if (DOTFORM) {
    if (tid < 128) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < DMAX; ++k) acc = fma(rs[tid * DMAX + k], rs[tid * DMAX + k], acc);
      rn[tid] = acc;
    }
    __syncthreads();
  }
  // interior tiles (the vast majority) need neither the padding rule nor the diagonal term: uniform per workgroup
  const bool edge = (r0 + 128 > nr) || (c0 + 128 > nc);
  const bool diag = same_set && (col_g0 + c0 < r0 + 128) && (r0 < col_g0 + c0 + 128);

  for (int rr = 0; rr < 32; ++rr) {
    const int i = w * 32 + rr;
    const int gi = r0 + i;
    double s0 = 0.0, s1 = 0.0;
    if (DOTFORM) {
#pragma unroll
      for (int k = 0; k < DMAX; ++k) {
        const double x = rs[i * DMAX + k];
        s0 = fma(x, za[k], s0);
        s1 = fma(x, zb[k], s1);
      }
      const double r = rn[i];
      s0 = fma(-2.0, s0, r + na);
      s1 = fma(-2.0, s1, r + nb);
    } else if (D != 0) {
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double x = rs[i * D + k];
        const double e0 = x - za[k], e1 = x - zb[k];
        s0 = fma(e0, e0, s0);
        s1 = fma(e1, e1, s1);
      }
    } else {
      for (int k = 0; k < d; ++k) {
        const double x = rs[i * d + k];
        const double e0 = x - cs[(2 * lane) * d + k], e1 = x - cs[(2 * lane + 1) * d + k];
        s0 = fma(e0, e0, s0);
        s1 = fma(e1, e1, s1);
      }
    }
    double v0, v1;
    if (MODE == 1) {
      v0 = s0;
      v1 = s1;
    } else if (MODE == 2) {  // timing experiment: stores only
      v0 = 1.0;
      v1 = 2.0;
    } else {
      v0 = exp2_neg(lsf - s0);
      v1 = exp2_neg(lsf - s1);
      if (diag) {
        if (gi == col_g0 + ca) v0 += p.diag_add;
        if (gi == col_g0 + cb) v1 += p.diag_add;
      }
    }
    if (edge) {
      const bool rok = gi < nr;
      if (!(rok && ca < nc)) v0 = (same_set && gi == col_g0 + ca) ? 1.0 : 0.0;  // padding: identity / zero
      if (!(rok && cb < nc)) v1 = (same_set && gi == col_g0 + cb) ? 1.0 : 0.0;
    }
    d2 o = {v0, v1};
    // streamed once, read back much later: non-temporal (SE build at N = 16384, d = 8: 0.26 -> 0.24 ms, profiles/r03_kbuild_bench.txt)
    __builtin_nontemporal_store(o, reinterpret_cast<d2*>(out + (long)gi * ld + ca));
  }
}

// ---- scalar-operand form (round 4).  The row point of an output row is the same for all 64 lanes of the wave that writes the
// row segment: it belongs in SGPRs, not in LDS.  A pre-pass stores the centred, scaled row points and their squared norms
// once (se_scale_rows_kernel: N x D + N doubles, 1 MiB at N = 16384); the tile kernel then fetches a row with ONE s_load_dwordx16
// and feeds it to v_fma_f64 as the scalar operand: no LDS, no barrier, no broadcast reads (the LDS form spends four
// ds_read_b128 + one ds_read_b64 per 128 output elements, and d = 8 costs 0.033 ms more than d = 4 for exactly that reason:
// profiles/r04_kbuild_bench.txt).  Same operations in the same order as se_build_kernel<D, 0>: the same bits (tests).
template <int D>
__global__ __launch_bounds__(256) void se_scale_rows_kernel(const double* __restrict__ R, int nr, int nr_pad, SEParams p,
                                                            const double* __restrict__ centre, double* __restrict__ Rs,
                                                            double* __restrict__ Rn) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nr_pad) return;
  const double pre = 0.8493218002880191;  // sqrt(log2(e) / 2)
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double v = (i < nr) ? (R[(long)i * D + k] - centre[k]) / p.ell[k] * pre : 0.0;
    Rs[(long)i * D + k] = v;
    acc = fma(v, v, acc);
  }
  Rn[i] = acc;
}

template <int D>
__global__ __launch_bounds__(256) void se_build_srow_kernel(const double* __restrict__ Rs, const double* __restrict__ Rn, int nr,
                                                            const double* __restrict__ Cs, const double* __restrict__ Cn, int nc,
                                                            SEParams p, int same_set, int lower_only, double* __restrict__ out,
                                                            long ld, int col_g0) {
  int tc = blockIdx.x, tr = blockIdx.y;
  if (lower_only == 2) {
    const int id = blockIdx.x;
    int t = (int)((sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
    while (t * (t + 1) / 2 > id) --t;
    while ((t + 1) * (t + 2) / 2 <= id) ++t;
    tr = t;
    tc = id - t * (t + 1) / 2;
  }
  if (lower_only == 1 && col_g0 + tc * 128 > tr * 128 + 127) return;
  const int r0 = tr * 128, c0 = tc * 128;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (one-wave workgroups, four per tile, measured the same)
  const double lsf = log2(p.sf2);
  const int ca = c0 + 2 * lane, cb = ca + 1;
  double za[D], zb[D];  // this lane's two column points (scaled by the pre-pass: zero rows beyond nc) and their norms
#pragma unroll
  for (int k = 0; k < D; ++k) {
    za[k] = Cs[(long)ca * D + k];
    zb[k] = Cs[(long)cb * D + k];
  }
  const double na = Cn[ca], nb = Cn[cb];
  const bool edge = (r0 + 128 > nr) || (c0 + 128 > nc);
  const bool diag = same_set && (col_g0 + c0 < r0 + 128) && (r0 < col_g0 + c0 + 128);
  const double* __restrict__ rp = Rs + (long)(r0 + w * 32) * D;  // wave-uniform: scalar loads
  const double* __restrict__ np = Rn + (r0 + w * 32);
#pragma unroll 4
  for (int rr = 0; rr < 32; ++rr) {
    const int gi = r0 + w * 32 + rr;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const double x = rp[rr * D + k];
      s0 = fma(x, za[k], s0);
      s1 = fma(x, zb[k], s1);
    }
    const double r = np[rr];
    s0 = fma(-2.0, s0, r + na);
    s1 = fma(-2.0, s1, r + nb);
    double v0 = exp2_neg(lsf - s0), v1 = exp2_neg(lsf - s1);
    if (diag) {
      if (gi == col_g0 + ca) v0 += p.diag_add;
      if (gi == col_g0 + cb) v1 += p.diag_add;
    }
    if (edge) {
      const bool rok = gi < nr;
      if (!(rok && ca < nc)) v0 = (same_set && gi == col_g0 + ca) ? 1.0 : 0.0;
      if (!(rok && cb < nc)) v1 = (same_set && gi == col_g0 + cb) ? 1.0 : 0.0;
    }
    d2 o = {v0, v1};
    __builtin_nontemporal_store(o, reinterpret_cast<d2*>(out + (long)gi * ld + ca));
  }
}

template <int MODE>
static void se_build_dispatch(int dd, dim3 grid, dim3 block, hipStream_t s, const double* R, int nr, const double* Cc, int nc,
                              const SEParams& p, int same_set, int lo, double* out, long ld, int col_g0,
                              const double* centre) {
#define GPK_SE_CASE(DD)                                                                                                 \
  case DD:                                                                                                              \
    hipLaunchKernelGGL((se_build_kernel<DD, MODE>), grid, block, 0, s, R, nr, Cc, nc, p, same_set, lo, out, ld, col_g0, centre); \
    break;
  switch (dd) {
    GPK_SE_CASE(1)
    GPK_SE_CASE(2)
    GPK_SE_CASE(3)
    GPK_SE_CASE(4)
    GPK_SE_CASE(5)
    GPK_SE_CASE(6)
    GPK_SE_CASE(7)
    GPK_SE_CASE(8)
    default:
      hipLaunchKernelGGL((se_build_kernel<0, MODE>), grid, block, 0, s, R, nr, Cc, nc, p, same_set, lo, out, ld, col_g0, centre);
  }
#undef GPK_SE_CASE
}

template <int D>
static void se_build_srow_launch(dim3 grid, hipStream_t s, const double* R, int nr, int nr_pad, const double* Cc, int nc, int nc_pad,
                                 const SEParams& p, int same_set, int lo, double* out, long ld, int col_g0, const double* centre,
                                 const SEScalarForm& f) {
  double* Rs = f.scratch;
  const double* Rn = (f.rows_ready && f.rows_norms) ? f.rows_norms : f.scratch + (size_t)nr_pad * D;
  const double *Cs, *Cn;
  if (!f.rows_ready) hipLaunchKernelGGL((se_scale_rows_kernel<D>), dim3((nr_pad + 255) / 256), dim3(256), 0, s, R, nr, nr_pad, p, centre, Rs, Rs + (size_t)nr_pad * D);
  // the column points: an aligned block of the row points (K, the block columns of a block-cyclic K), a set an earlier build has
  // scaled (Kxz after K: the training points), or scaled here behind the rows
  const bool cols_in_rows = Cc >= R && Cc + (size_t)nc * D <= R + (size_t)nr * D && (Cc - R) % D == 0 && (Cc - R) / D + nc_pad <= nr_pad;
  if (cols_in_rows) {
    Cs = Rs + (Cc - R);
    Cn = Rn + (Cc - R) / D;
  } else if (f.cols_scaled) {
    Cs = f.cols_scaled;
    Cn = f.cols_scaled + (size_t)f.cols_pad * D;
  } else {
    double* cs = f.scratch + (size_t)nr_pad * (D + 1);
    hipLaunchKernelGGL((se_scale_rows_kernel<D>), dim3((nc_pad + 255) / 256), dim3(256), 0, s, Cc, nc, nc_pad, p, centre, cs, cs + (size_t)nc_pad * D);
    Cs = cs;
    Cn = cs + (size_t)nc_pad * D;
  }
  hipLaunchKernelGGL((se_build_srow_kernel<D>), grid, dim3(256), 0, s, Rs, Rn, nr, Cs, Cn, nc, p, same_set, lo, out, ld, col_g0);
}

int launch_se_scale(const double* P, int n, int n_pad, const SEParams& p, const double* centre, double* S, hipStream_t s) {
  if (p.d < 1 || p.d > 8 || !S) return gpk_set_err(-1, "se_scale: d=%d outside 1..8 or no scratch", p.d);
  const dim3 grid((n_pad + 255) / 256), block(256);
  switch (p.d) {
#define GPK_SCALE_CASE(DD) case DD: hipLaunchKernelGGL((se_scale_rows_kernel<DD>), grid, block, 0, s, P, n, n_pad, p, centre, S, S + (size_t)n_pad * DD); break;
    GPK_SCALE_CASE(1) GPK_SCALE_CASE(2) GPK_SCALE_CASE(3) GPK_SCALE_CASE(4) GPK_SCALE_CASE(5) GPK_SCALE_CASE(6) GPK_SCALE_CASE(7) GPK_SCALE_CASE(8)
#undef GPK_SCALE_CASE
  }
  GPK_HIP(hipGetLastError());
  return 0;
}

int launch_se_build(const double* R, int nr, int nr_pad, const double* Cc, int nc, int nc_pad, const SEParams& p,
                    int same_set, int mode, double* out, long ld, long lower_only, hipStream_t s, int col_g0, const double* centre,
                    const SEScalarForm& form) {
  if (!centre) centre = Cc;  // the call's first column point
  if (nr_pad % 128 || nc_pad % 128 || (ld & 1) || ((uintptr_t)out & 15))
    return gpk_set_err(-1, "se_build: padded shape %dx%d ld %ld must be tile aligned", nr_pad, nc_pad, ld);
  if (p.d < 1 || p.d > 32) return gpk_set_err(-1, "se_build: input dimension d=%d outside 1..32", p.d);
  dim3 grid(nc_pad / 128, nr_pad / 128), block(256);
  int lo = (int)lower_only;
  if (lo && col_g0 == 0 && nr_pad == nc_pad) {  // square lower build: triangular 1-D grid
    const unsigned nt = nr_pad / 128;
    grid = dim3(nt * (nt + 1) / 2, 1);
    lo = 2;
  }
  if (mode == 0 && form.scratch && p.d <= 8) {  // scalar-operand form
    if (form.cols_scaled && form.cols_pad < nc_pad) return gpk_set_err(-1, "se_build: %d scaled column points, %d needed", form.cols_pad, nc_pad);
    switch (p.d) {
#define GPK_SROW_CASE(DD) case DD: se_build_srow_launch<DD>(grid, s, R, nr, nr_pad, Cc, nc, nc_pad, p, same_set, lo, out, ld, col_g0, centre, form); break;
      GPK_SROW_CASE(1) GPK_SROW_CASE(2) GPK_SROW_CASE(3) GPK_SROW_CASE(4) GPK_SROW_CASE(5) GPK_SROW_CASE(6) GPK_SROW_CASE(7) GPK_SROW_CASE(8)
#undef GPK_SROW_CASE
    }
    GPK_HIP(hipGetLastError());
    return 0;
  }
  if (mode == 0) se_build_dispatch<0>(p.d, grid, block, s, R, nr, Cc, nc, p, same_set, lo, out, ld, col_g0, centre);
  else if (mode == 1) se_build_dispatch<1>(p.d, grid, block, s, R, nr, Cc, nc, p, same_set, lo, out, ld, col_g0, centre);
  else se_build_dispatch<2>(p.d, grid, block, s, R, nr, Cc, nc, p, same_set, lo, out, ld, col_g0, centre);
  GPK_HIP(hipGetLastError());
  return 0;
}
