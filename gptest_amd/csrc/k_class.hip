// Binary GP classification, Laplace approximation: element-wise pieces of Rasmussen & Williams Alg. 3.1 / 3.2 on the
// device.  SURVEY.md 8(f) item 3: what the reference's GPc.py (a non-parsing sketch: NormCDF.logpyf GPc.py:5-6,
// logistic_function GPc.py:13-14, "Rasmussen eq. 3.12" GPc.py:40-46) and GP_classification_demo.py were heading to.
// The O(n^3) part -- B = I + W^1/2 K W^1/2 and its Cholesky -- runs on the same MFMA factorisation as regression.
#include "gpk_internal.h"

// log Phi(z) and r(z) = phi(z)/Phi(z), stable for very negative z through erfcx:  Phi(z) = erfcx(t) exp(-t^2) / 2,  t = -z/sqrt2
__device__ inline void probit_terms(double z, double* logphi, double* ratio) {
  if (z > -5.0) {
    const double p = normcdf(z);
    *logphi = log(p);
    *ratio = exp(-0.5 * z * z) * 0.3989422804014327 / p;
  } else {
    const double t = -z * 0.7071067811865476;
    const double e = erfcx(t);
    *logphi = log(0.5 * e) - t * t;
    *ratio = 0.7978845608028654 / e;  // sqrt(2/pi) / erfcx(t)
  }
}

// y in {-1, +1}.  link 0 (probit): log p = log Phi(y f), g = y r, w = r^2 + z r (z = y f).
//                 link 1 (logit):  log p = -log(1 + exp(-y f)), g = t - pi, w = pi (1 - pi), pi = 1/(1+exp(-f)), t = (y+1)/2.
__global__ void class_stats_kernel(const double* __restrict__ y, const double* __restrict__ f, int n, int link,
                                   double* __restrict__ g, double* __restrict__ w, double* __restrict__ sw,
                                   double* __restrict__ logp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double yy = y[i], ff = f[i];
  double gi, wi, lp;
  if (link == 0) {
    const double z = yy * ff;
    double r;
    probit_terms(z, &lp, &r);
    gi = yy * r;
    wi = r * r + z * r;
  } else {
    const double z = yy * ff;
    lp = (z > 0.0) ? -log1p(exp(-z)) : z - log1p(exp(z));
    const double pi = (ff > 0.0) ? 1.0 / (1.0 + exp(-ff)) : exp(ff) / (1.0 + exp(ff));
    gi = 0.5 * (yy + 1.0) - pi;
    wi = pi * (1.0 - pi);
  }
  if (wi < 0.0) wi = 0.0;  // both likelihoods are log-concave; guards rounding at extreme z
  if (g) g[i] = gi;
  if (w) w[i] = wi;
  if (sw) sw[i] = sqrt(wi);
  if (logp) logp[i] = lp;
}

// out = a*x + y   element-wise products:  mode 0: out = p .* q + r ;  mode 1: out = r - p .* q ;  mode 2: out = p .* q
__global__ void class_ew_kernel(const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ r,
                                int n, int mode, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double pq = p[i] * q[i];
  out[i] = (mode == 0) ? pq + r[i] : (mode == 1) ? r[i] - pq : pq;
}

// U[j][i] = R[j][i] * sw[i] for j < rows, i < n; zero on the padding (rows_pad x npad)
__global__ void class_scale_rows_kernel(const double* __restrict__ R, long ldr, const double* __restrict__ sw, int n,
                                        int rows, int npad, double* __restrict__ U, long ldu) {
  const int j = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += gridDim.x * blockDim.x)
    U[(long)j * ldu + i] = (j < rows && i < n) ? R[(long)j * ldr + i] * sw[i] : 0.0;
}

// averaged predictive probability of class +1:  probit: Phi(mean / sqrt(1 + var)) (R&W eq. 3.82, exact);
// logit: the probit approximation of the logistic integral, sigma(mean / sqrt(1 + pi var / 8)) (MacKay 1992)
__global__ void class_prob_kernel(const double* __restrict__ mean, const double* __restrict__ var, int n, int link,
                                  double* __restrict__ prob) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = fmax(var[i], 0.0);
  if (link == 0) {
    prob[i] = normcdf(mean[i] / sqrt(1.0 + v));
  } else {
    const double k = mean[i] / sqrt(1.0 + 0.39269908169872414 * v);
    prob[i] = (k > 0.0) ? 1.0 / (1.0 + exp(-k)) : exp(k) / (1.0 + exp(k));
  }
}

#define CLS_GRID(n) dim3(((n) + 255) / 256), dim3(256)

int launch_class_stats(const double* y, const double* f, int n, int link, double* g, double* w, double* sw, double* logp,
                       hipStream_t s) {
  hipLaunchKernelGGL(class_stats_kernel, CLS_GRID(n), 0, s, y, f, n, link, g, w, sw, logp);
  GPK_HIP(hipGetLastError());
  return 0;
}
int launch_class_ew(const double* p, const double* q, const double* r, int n, int mode, double* out, hipStream_t s) {
  hipLaunchKernelGGL(class_ew_kernel, CLS_GRID(n), 0, s, p, q, r, n, mode, out);
  GPK_HIP(hipGetLastError());
  return 0;
}
int launch_class_scale_rows(const double* R, long ldr, const double* sw, int n, int rows, int rows_pad, int npad, double* U,
                            long ldu, hipStream_t s) {
  const int gx = (npad + 255) / 256 > 16 ? 16 : (npad + 255) / 256;
  hipLaunchKernelGGL(class_scale_rows_kernel, dim3(gx, rows_pad), dim3(256), 0, s, R, ldr, sw, n, rows, npad, U, ldu);
  GPK_HIP(hipGetLastError());
  return 0;
}
int launch_class_prob(const double* mean, const double* var, int n, int link, double* prob, hipStream_t s) {
  hipLaunchKernelGGL(class_prob_kernel, CLS_GRID(n), 0, s, mean, var, n, link, prob);
  GPK_HIP(hipGetLastError());
  return 0;
}
