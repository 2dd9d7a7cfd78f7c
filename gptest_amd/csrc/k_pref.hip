// Preference-GP (probit pairs) Laplace mode finding on the device.
// Replaces PrefProbit.derivatives (the Python loop over pairs that builds the dense W, GPpref.py:68-88)
// and the Newton loop of PreferenceGaussianProcess.calc_laplace (GPpref.py:137-157).
//
// With d_p = e_v - e_u per pair:  z_p = y_p d_p.f /(sqrt2 sigma),  W = sum_p c_p d_p d_p^T (c >= 0: the
// probit likelihood is log-concave),  grad = sum_p q_p d_p.  The reference forms inv(K^-1 + W) densely
// (2 nx^3 per iteration).  Here  (K^-1 + D^T C D)^-1 b = K b - K D^T C^1/2 B^-1 C^1/2 D K b  with
// B = I + C^1/2 (D K D^T) C^1/2  (npairs x npairs, SPD, eigenvalues >= 1): K is only ever multiplied, the
// per-iteration factorisation is of B, on the same MFMA Cholesky as everything else.
#include "gpk_internal.h"

// z, q = y*isq*N/Phi, c = i2var*(z*N/Phi + (N/Phi)^2), logphi = log Phi(z)      (GPpref.py:56-58,69-80)
__global__ void pref_pair_stats_kernel(const long long* __restrict__ uvi, const double* __restrict__ y,
                                       const double* __restrict__ f, int npairs, double isq, double* z_out,
                                       double* q_out, double* c_out, double* logphi_out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  const long long u = uvi[2 * p], v = uvi[2 * p + 1];
  const double yy = y[p];
  const double z = yy * (isq * (f[v] - f[u]));
  const double phi = normcdf(z);
  const double zz = fmin(fmax(z, -1e150), 1e150);
  const double nz = exp(-0.5 * zz * zz) * 0.3989422804014327;  // 1/sqrt(2 pi)
  const double ratio = nz / phi;
  if (z_out) z_out[p] = z;
  if (q_out) q_out[p] = yy * isq * ratio;
  if (c_out) c_out[p] = isq * isq * (z * ratio + ratio * ratio);
  if (logphi_out) logphi_out[p] = log(phi);
}

// t_p = c_p * (f[v]-f[u]) + q_p      ->  b = D^T t = W f + grad
__global__ void pref_pair_t_kernel(const long long* __restrict__ uvi, const double* __restrict__ f,
                                   const double* __restrict__ c, const double* __restrict__ q, int npairs, double* t) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  t[p] = c[p] * (f[uvi[2 * p + 1]] - f[uvi[2 * p]]) + q[p];
}

// out += D^T (scale .* t):  out[v] += s_p t_p,  out[u] -= s_p t_p   (scale may be null; sqrt applied if do_sqrt)
__global__ void pref_scatter_kernel(const long long* __restrict__ uvi, const double* __restrict__ t,
                                    const double* __restrict__ scale, int do_sqrt, int npairs, double* out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  double s = t[p];
  if (scale) s *= do_sqrt ? sqrt(scale[p]) : scale[p];
  atomicAdd(out + uvi[2 * p + 1], s);  // duplicate-free pair sets (GP_preference_demo.py:40): no contention
  atomicAdd(out + uvi[2 * p], -s);
}

// r_p = sqrt(c_p) * (x[v]-x[u])
__global__ void pref_gather_kernel(const long long* __restrict__ uvi, const double* __restrict__ x,
                                   const double* __restrict__ c, int npairs, double* r) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  r[p] = sqrt(c[p]) * (x[uvi[2 * p + 1]] - x[uvi[2 * p]]);
}

// U[j][p] = sqrt(c_p) * (R[j][v_p] - R[j][u_p]) for j < rows, p < npairs; zero on the padding (rows_pad x npad)
__global__ void pref_gather_rows_kernel(const long long* __restrict__ uvi, const double* __restrict__ R, long ldr,
                                        const double* __restrict__ c, int npairs, int rows, int npad,
                                        double* __restrict__ U, long ldu) {
  const int j = blockIdx.y;
  const double* Rj = R + (long)j * ldr;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npad; p += gridDim.x * blockDim.x) {
    double v = 0.0;
    if (j < rows && p < npairs) v = sqrt(c[p]) * (Rj[uvi[2 * p + 1]] - Rj[uvi[2 * p]]);
    U[(long)j * ldu + p] = v;
  }
}

// S[p][q] = K[v_p,v_q] - K[v_p,u_q] - K[u_p,v_q] + K[u_p,u_q]   (D K D^T; K symmetric, full copy)
__global__ void pref_build_S_kernel(const double* __restrict__ K, long ldk, const long long* __restrict__ uvi,
                                    int npairs, double* __restrict__ S, long lds_) {
  const int p = blockIdx.y;
  const long long up = uvi[2 * p], vp = uvi[2 * p + 1];
  const double* Ku = K + up * ldk;
  const double* Kv = K + vp * ldk;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < npairs; q += gridDim.x * blockDim.x) {
    const long long uq = uvi[2 * q], vq = uvi[2 * q + 1];
    S[(long)p * lds_ + q] = Kv[vq] - Kv[uq] - Ku[vq] + Ku[uq];
  }
}

// B = I + diag(sqrt c) S diag(sqrt c) on the lower tiles; identity on the padding
__global__ void pref_build_B_kernel(const double* __restrict__ S, long lds_, const double* __restrict__ c, int npairs,
                                    int npad, double* __restrict__ B, long ldb) {
  const int p = blockIdx.y;
  const double sp = (p < npairs) ? sqrt(c[p]) : 0.0;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < npad; q += gridDim.x * blockDim.x) {
    double v = (p == q) ? 1.0 : 0.0;
    if (p < npairs && q < npairs) v += sp * S[(long)p * lds_ + q] * sqrt(c[q]);
    B[(long)p * ldb + q] = v;
  }
}

// out[0] = max_i |a_i - b_i|   (single workgroup, deterministic)
__global__ void maxabsdiff_kernel(const double* __restrict__ a, const double* __restrict__ b, int n, double* out) {
  __shared__ double sm[16];
  double m = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmax(m, fabs(a[i] - b[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r = fmax(r, sm[w]);
    *out = r;
  }
}
__global__ void sum_kernel(const double* __restrict__ a, int n, double* out) {
  __shared__ double sm[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += a[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sm[w];
    *out = r;
  }
}
__global__ void axpby_kernel(const double* __restrict__ a, const double* __restrict__ b, int n, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] - b[i];
}

#define PREF_GRID(n) dim3(((n) + 255) / 256), dim3(256)

int launch_pref_pair_stats(const long long* uvi, const double* y, const double* f, int npairs, double isq, double* z,
                           double* q, double* c, double* logphi, hipStream_t s) {
  hipLaunchKernelGGL(pref_pair_stats_kernel, PREF_GRID(npairs), 0, s, uvi, y, f, npairs, isq, z, q, c, logphi);
  GPK_HIP(hipGetLastError());
  return 0;
}
int launch_pref_pair_t(const long long* uvi, const double* f, const double* c, const double* q, int npairs, double* t,
                       hipStream_t s) {
  hipLaunchKernelGGL(pref_pair_t_kernel, PREF_GRID(npairs), 0, s, uvi, f, c, q, npairs, t);
  GPK_HIP(hipGetLastError());
  return 0;
}
int launch_pref_scatter(const long long* uvi, const double* t, const double* scale, int do_sqrt, int npairs, double* out,
                        hipStream_t s) {
  hipLaunchKernelGGL(pref_scatter_kernel, PREF_GRID(npairs), 0, s, uvi, t, scale, do_sqrt, npairs, out);
  GPK_HIP(hipGetLastError());
  return 0;
}
int launch_pref_gather(const long long* uvi, const double* x, const double* c, int npairs, double* r, hipStream_t s) {
  hipLaunchKernelGGL(pref_gather_kernel, PREF_GRID(npairs), 0, s, uvi, x, c, npairs, r);
  GPK_HIP(hipGetLastError());
  return 0;
}
int launch_pref_gather_rows(const long long* uvi, const double* R, long ldr, const double* c, int npairs, int rows,
                             int rows_pad, int npad, double* U, long ldu, hipStream_t s) {
  const int gx = (npad + 255) / 256 > 16 ? 16 : (npad + 255) / 256;
  hipLaunchKernelGGL(pref_gather_rows_kernel, dim3(gx, rows_pad), dim3(256), 0, s, uvi, R, ldr, c, npairs, rows, npad, U, ldu);
  GPK_HIP(hipGetLastError());
  return 0;
}
int launch_pref_build_S(const double* K, long ldk, const long long* uvi, int npairs, double* S, long lds_, hipStream_t s) {
  hipLaunchKernelGGL(pref_build_S_kernel, dim3((npairs + 255) / 256 > 16 ? 16 : (npairs + 255) / 256, npairs), dim3(256), 0, s,
                     K, ldk, uvi, npairs, S, lds_);
  GPK_HIP(hipGetLastError());
  return 0;
}
int launch_pref_build_B(const double* S, long lds_, const double* c, int npairs, int npad, double* B, long ldb,
                        hipStream_t s) {
  hipLaunchKernelGGL(pref_build_B_kernel, dim3((npad + 255) / 256 > 16 ? 16 : (npad + 255) / 256, npad), dim3(256), 0, s, S,
                     lds_, c, npairs, npad, B, ldb);
  GPK_HIP(hipGetLastError());
  return 0;
}
int launch_maxabsdiff(const double* a, const double* b, int n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(maxabsdiff_kernel, dim3(1), dim3(1024), 0, s, a, b, n, out);
  GPK_HIP(hipGetLastError());
  return 0;
}
int launch_sum(const double* a, int n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, s, a, n, out);
  GPK_HIP(hipGetLastError());
  return 0;
}
int launch_sub(const double* a, const double* b, int n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(axpby_kernel, PREF_GRID(n), 0, s, a, b, n, out);
  GPK_HIP(hipGetLastError());
  return 0;
}
