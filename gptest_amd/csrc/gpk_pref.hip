// Host driver of the preference-GP Laplace loop (PreferenceGaussianProcess.calc_laplace, GPpref.py:112-157)
// on top of a fitted model handle (K + eps*I factored, unfactored copy kept).  Everything stays in HBM;
// per Newton iteration the host sees one scalar (max |f_new - f|).
#include "../../include/gpk.h"
#include "../../include/gpk_bench.h"
#include "gpk_core.h"
#include <cmath>

// PrefProbit's per-pair quantities (GPpref.py:56-58, 68-80, 90-92) for host callers of the likelihood object itself:
// z_p = y_p (f_v - f_u) / (sqrt2 sigma), q_p = y_p N(z_p) / (sqrt2 sigma Phi(z_p)) (what derivatives() scatters into dpy_df
// with -/+ signs), c_p = (z_p N/Phi + (N/Phi)^2) / (2 sigma^2) (minus the `inner` of GPpref.py:80: W's four entries per pair
// are +c, +c, -c, -c) and log Phi(z_p).  The same kernel the Laplace loop runs every Newton iteration.  No model handle.
extern "C" int gpk_pref_pair_stats(const long long* uvi, const double* y, int npairs, double sigma, const double* f, int nx,
                                   double* z_out, double* q_out, double* c_out, double* logphi_out) {
  GPK_TRY(rt_require());
  if (!uvi || !y || !f) return gpk_set_err(-1, "gpk_pref_pair_stats: null pointer");
  if (npairs < 1 || nx < 1 || !(sigma > 0.0)) return gpk_set_err(-1, "gpk_pref_pair_stats: bad arguments");
  for (int i = 0; i < 2 * npairs; ++i)
    if (uvi[i] < 0 || uvi[i] >= nx) return gpk_set_err(-1, "gpk_pref_pair_stats: pair index %lld outside 0..%d", uvi[i], nx - 1);
  hipStream_t s = G.s_main;
  Scratch sc;
  long long* duvi = nullptr;
  double *dy = nullptr, *df = nullptr, *dout = nullptr;
  GPK_TRY(sc.get(&duvi, (size_t)2 * npairs));
  GPK_TRY(sc.get(&dy, (size_t)npairs));
  GPK_TRY(sc.get(&df, (size_t)nx));
  GPK_TRY(sc.get(&dout, (size_t)4 * npairs));
  GPK_HIP(hipMemcpyAsync(duvi, uvi, (size_t)2 * npairs * 8, hipMemcpyHostToDevice, s));
  GPK_HIP(hipMemcpyAsync(dy, y, (size_t)npairs * 8, hipMemcpyHostToDevice, s));
  GPK_HIP(hipMemcpyAsync(df, f, (size_t)nx * 8, hipMemcpyHostToDevice, s));
  GPK_TRY(launch_pref_pair_stats(duvi, dy, df, npairs, 1.0 / (sigma * std::sqrt(2.0)), dout, dout + npairs, dout + 2 * (size_t)npairs,
                                 dout + 3 * (size_t)npairs, s));
  double* outs[4] = {z_out, q_out, c_out, logphi_out};
  for (int k = 0; k < 4; ++k)
    if (outs[k]) GPK_HIP(hipMemcpyAsync(outs[k], dout + (size_t)k * npairs, (size_t)npairs * 8, hipMemcpyDeviceToHost, s));
  GPK_HIP(hipStreamSynchronize(s));
  return 0;
}

extern "C" int gpk_pref_laplace(gpk_model* m, const long long* uvi, const double* y, int npairs, double sigma,
                                double* f_inout, double delta_f, int max_iter, double* lml_out, int* iters_out,
                                double* ferr_trace) {
  GPK_TRY(rt_require());
  if (!m || !uvi || !y || !f_inout) return gpk_set_err(-1, "gpk_pref_laplace: null pointer");
  if (!m->dKcopy || !m->factored) return gpk_set_err(-1, "gpk_pref_laplace: fit the model with the keep-K flag (and a factorisation)");
  if (m->P != 1) return gpk_set_err(-1, "gpk_pref_laplace: single-GPU models only");
  if (npairs < 1 || max_iter < 1 || !(sigma > 0.0)) return gpk_set_err(-1, "gpk_pref_laplace: bad arguments");
  const int nx = m->N, np_ = m->Npad;
  for (int i = 0; i < 2 * npairs; ++i)
    if (uvi[i] < 0 || uvi[i] >= nx) return gpk_set_err(-1, "gpk_pref_laplace: pair index %lld outside 0..%d", uvi[i], nx - 1);
  hipStream_t s = G.s_main;
  gpk_model* mb = nullptr;  // the npairs x npairs system B = I + C^1/2 (D K D^T) C^1/2
  GPK_TRY(model_create(&mb, npairs, 1, false, 1, 0));
  const int nbp = mb->Npad;
  long long* duvi = nullptr;
  double *dy = nullptr, *df = nullptr, *dfn = nullptr, *dq = nullptr, *dc = nullptr, *dlp = nullptr, *dt = nullptr,
         *dr = nullptr, *db = nullptr, *dkb = nullptr, *dsv = nullptr, *dks = nullptr, *dS = nullptr, *dscal = nullptr;
  int rc = 0;
  rc |= dev_alloc_t(&duvi, (size_t)2 * npairs);
  rc |= dev_alloc_t(&dy, (size_t)npairs);
  rc |= dev_alloc_t(&df, (size_t)np_);
  rc |= dev_alloc_t(&dfn, (size_t)np_);
  rc |= dev_alloc_t(&dq, (size_t)npairs);
  rc |= dev_alloc_t(&dc, (size_t)npairs);
  rc |= dev_alloc_t(&dlp, (size_t)npairs);
  rc |= dev_alloc_t(&dt, (size_t)npairs);
  rc |= dev_alloc_t(&dr, (size_t)nbp);
  rc |= dev_alloc_t(&db, (size_t)np_);
  rc |= dev_alloc_t(&dkb, (size_t)np_);
  rc |= dev_alloc_t(&dsv, (size_t)np_);
  rc |= dev_alloc_t(&dks, (size_t)np_);
  rc |= dev_alloc_t(&dS, (size_t)npairs * nbp);
  rc |= dev_alloc_t(&dscal, 8);
  const double isq = 1.0 / (sigma * std::sqrt(2.0));
  int iters = 0;
  double lml = 0.0;
#define PREF_HIP(x)                                                                         \
  if (!rc && (x) != hipSuccess) rc = gpk_set_err(-2, "%s failed: %s", #x, hipGetErrorString(hipGetLastError()))
#define PREF_DO(x) \
  if (!rc) rc = (x)
  PREF_HIP(hipMemcpyAsync(duvi, uvi, (size_t)2 * npairs * 8, hipMemcpyHostToDevice, s));
  PREF_HIP(hipMemcpyAsync(dy, y, (size_t)npairs * 8, hipMemcpyHostToDevice, s));
  PREF_HIP(hipMemsetAsync(df, 0, (size_t)np_ * 8, s));
  PREF_HIP(hipMemcpyAsync(df, f_inout, (size_t)nx * 8, hipMemcpyHostToDevice, s));
  PREF_HIP(hipMemsetAsync(dfn, 0, (size_t)np_ * 8, s));
  PREF_DO(launch_pref_build_S(m->dKcopy, np_, duvi, npairs, dS, nbp, s));
  for (int it = 0; it < max_iter && !rc; ++it) {
    PREF_DO(launch_pref_pair_stats(duvi, dy, df, npairs, isq, nullptr, dq, dc, nullptr, s));
    PREF_DO(launch_pref_pair_t(duvi, df, dc, dq, npairs, dt, s));
    PREF_HIP(hipMemsetAsync(db, 0, (size_t)np_ * 8, s));
    PREF_DO(launch_pref_scatter(duvi, dt, nullptr, 0, npairs, db, s));            // b = W f + grad
    PREF_DO(launch_gemv(m->dKcopy, np_, np_, np_, db, dkb, 1.0, 0.0, s));          // K b
    PREF_HIP(hipMemsetAsync(dr, 0, (size_t)nbp * 8, s));
    PREF_DO(launch_pref_gather(duvi, dkb, dc, npairs, dr, s));                     // C^1/2 D K b
    PREF_DO(launch_pref_build_B(dS, nbp, dc, npairs, nbp, mb->dK, mb->ldk, s));
    PREF_DO(potrf_forward(mb, dr));                                               // factor B, L_B^-1 (...) riding on it
    PREF_DO(trsv_backward(mb, dr));                                               // B^-1 (...)
    PREF_HIP(hipMemsetAsync(dsv, 0, (size_t)np_ * 8, s));
    PREF_DO(launch_pref_scatter(duvi, dr, dc, 1, npairs, dsv, s));                 // D^T C^1/2 x
    PREF_DO(launch_gemv(m->dKcopy, np_, np_, np_, dsv, dks, 1.0, 0.0, s));
    PREF_DO(launch_sub(dkb, dks, np_, dfn, s));                                    // f_new
    PREF_DO(launch_maxabsdiff(dfn, df, nx, dscal, s));
    double ferr = 0.0;
    int binfo = 0;
    PREF_HIP(hipMemcpyAsync(&ferr, dscal, 8, hipMemcpyDeviceToHost, s));
    PREF_HIP(hipMemcpyAsync(&binfo, mb->dinfo, 4, hipMemcpyDeviceToHost, s));
    PREF_HIP(hipStreamSynchronize(s));
    if (!rc) rc = info_timeout_rc(binfo);
    if (!rc && binfo > 0) rc = gpk_set_err(-4, "Newton system lost positive definiteness (minor %d)", binfo);
    if (rc) break;
    std::swap(df, dfn);
    iters = it + 1;
    if (ferr_trace) ferr_trace[it] = ferr;
    if (!(ferr > delta_f)) break;  // GPpref.py:140: while f_error > delta_f
  }
  // lml at the final f (GPpref.py:90-94, 144): sum log Phi(z) - f^T iK f / 2 - logdetK/2 - nx/2 log 2 pi
  double hs[2] = {0, 0};
  PREF_DO(launch_pref_pair_stats(duvi, dy, df, npairs, isq, nullptr, nullptr, nullptr, dlp, s));
  PREF_DO(launch_sum(dlp, npairs, dscal, s));
  PREF_HIP(hipMemcpyAsync(dks, df, (size_t)np_ * 8, hipMemcpyDeviceToDevice, s));
  PREF_DO(trsv_forward(m, dks));
  PREF_DO(launch_dot(dks, dks, np_, dscal + 1, s));
  PREF_HIP(hipMemcpyAsync(hs, dscal, 16, hipMemcpyDeviceToHost, s));
  PREF_HIP(hipMemcpyAsync(f_inout, df, (size_t)nx * 8, hipMemcpyDeviceToHost, s));
  PREF_HIP(hipStreamSynchronize(s));
  if (!rc) lml = hs[0] - 0.5 * hs[1] - 0.5 * m->logdet - 0.5 * nx * std::log(2.0 * M_PI);
  if (lml_out) *lml_out = lml;
  if (iters_out) *iters_out = iters;
#undef PREF_HIP
#undef PREF_DO
  dev_free(duvi, (size_t)2 * npairs * 8);
  dev_free(dy, (size_t)npairs * 8);
  dev_free(df, (size_t)np_ * 8);
  dev_free(dfn, (size_t)np_ * 8);
  dev_free(dq, (size_t)npairs * 8);
  dev_free(dc, (size_t)npairs * 8);
  dev_free(dlp, (size_t)npairs * 8);
  dev_free(dt, (size_t)npairs * 8);
  dev_free(dr, (size_t)nbp * 8);
  dev_free(db, (size_t)np_ * 8);
  dev_free(dkb, (size_t)np_ * 8);
  dev_free(dsv, (size_t)np_ * 8);
  dev_free(dks, (size_t)np_ * 8);
  dev_free(dS, (size_t)npairs * nbp * 8);
  dev_free(dscal, 64);
  model_destroy(mb);
  return rc;
}

// Latent posterior at test inputs under the Laplace approximation (Rasmussen & Williams eq. 3.21/3.24; the
// reference stops at the mode, GPpref.py:157 -- SURVEY 8(f) item 2 asks for this as the next step):
//   mean* = k*^T Ke^-1 f,   var* = k** - k*^T (Ke + W^-1)^-1 k* = k** - u^T B^-1 u,  u = C^1/2 D k*,
// with Ke = K + eps I as factored by gpk_fit, W = D^T C D at f, B = I + C^1/2 (D Ke D^T) C^1/2: the same
// npairs x npairs system as one Newton step, so K is never inverted here either.
extern "C" int gpk_pref_predict(gpk_model* m, const long long* uvi, const double* y, int npairs, double sigma,
                                const double* f_hat, const double* Z, int M, double* mean_out, double* var_out) {
  GPK_TRY(rt_require());
  if (!m || !uvi || !y || !f_hat || !Z || !mean_out || !var_out) return gpk_set_err(-1, "gpk_pref_predict: null pointer");
  if (!m->dKcopy || !m->factored) return gpk_set_err(-1, "gpk_pref_predict: fit the model with the keep-K flag (and a factorisation)");
  if (m->P != 1) return gpk_set_err(-1, "gpk_pref_predict: single-GPU models only");
  if (npairs < 1 || M < 1 || !(sigma > 0.0)) return gpk_set_err(-1, "gpk_pref_predict: bad arguments");
  const int nx = m->N, np_ = m->Npad, d = m->d;
  for (int i = 0; i < 2 * npairs; ++i)
    if (uvi[i] < 0 || uvi[i] >= nx) return gpk_set_err(-1, "gpk_pref_predict: pair index %lld outside 0..%d", uvi[i], nx - 1);
  hipStream_t s = G.s_main;
  gpk_model* mb = nullptr;
  GPK_TRY(model_create(&mb, npairs, 1, false, 1, 0));
  const int nbp = mb->Npad, Mpad = round_up(M, 128);
  long long* duvi = nullptr;
  double *dy = nullptr, *df = nullptr, *dc = nullptr, *dS = nullptr, *dZ = nullptr, *dR = nullptr, *dU = nullptr, *dV = nullptr,
         *dzero = nullptr, *dmean = nullptr, *dvar = nullptr, *dtmp = nullptr;
  int rc = 0;
  rc |= dev_alloc_t(&duvi, (size_t)2 * npairs);
  rc |= dev_alloc_t(&dy, (size_t)npairs);
  rc |= dev_alloc_t(&df, (size_t)np_);
  rc |= dev_alloc_t(&dc, (size_t)npairs);
  rc |= dev_alloc_t(&dS, (size_t)npairs * nbp);
  rc |= dev_alloc_t(&dZ, (size_t)M * d);
  rc |= dev_alloc_t(&dR, (size_t)Mpad * np_);
  rc |= dev_alloc_t(&dU, (size_t)Mpad * nbp);
  rc |= dev_alloc_t(&dV, (size_t)Mpad * nbp);
  rc |= dev_alloc_t(&dzero, (size_t)nbp);
  rc |= dev_alloc_t(&dmean, (size_t)Mpad);
  rc |= dev_alloc_t(&dvar, (size_t)Mpad);
  rc |= dev_alloc_t(&dtmp, (size_t)Mpad);
  const double isq = 1.0 / (sigma * std::sqrt(2.0));
#define PREF_HIP(x)                                                                         \
  if (!rc && (x) != hipSuccess) rc = gpk_set_err(-2, "%s failed: %s", #x, hipGetErrorString(hipGetLastError()))
#define PREF_DO(x) \
  if (!rc) rc = (x)
  PREF_HIP(hipMemcpyAsync(duvi, uvi, (size_t)2 * npairs * 8, hipMemcpyHostToDevice, s));
  PREF_HIP(hipMemcpyAsync(dy, y, (size_t)npairs * 8, hipMemcpyHostToDevice, s));
  PREF_HIP(hipMemsetAsync(df, 0, (size_t)np_ * 8, s));
  PREF_HIP(hipMemcpyAsync(df, f_hat, (size_t)nx * 8, hipMemcpyHostToDevice, s));
  PREF_HIP(hipMemcpyAsync(dZ, Z, (size_t)M * d * 8, hipMemcpyHostToDevice, s));
  PREF_HIP(hipMemsetAsync(dzero, 0, (size_t)nbp * 8, s));
  // B at f and its factor
  PREF_DO(launch_pref_pair_stats(duvi, dy, df, npairs, isq, nullptr, nullptr, dc, nullptr, s));
  PREF_DO(launch_pref_build_S(m->dKcopy, np_, duvi, npairs, dS, nbp, s));
  PREF_DO(launch_pref_build_B(dS, nbp, dc, npairs, nbp, mb->dK, mb->ldk, s));
  PREF_DO(potrf_device(mb));
  // k* rows, mean* = Kzx (Ke^-1 f)
  PREF_DO(launch_se_build(dZ, M, Mpad, m->dX, nx, np_, m->se, 0, 0, dR, np_, 0, s));
  PREF_DO(trsv_forward(m, df));
  PREF_DO(trsv_backward(m, df));
  PREF_DO(launch_gemv(dR, np_, M, np_, df, dmean, 1.0, 0.0, s));
  // u rows, V = U Lb^-T, var* = sf2 - |V_j|^2
  PREF_DO(launch_pref_gather_rows(duvi, dR, np_, dc, npairs, M, Mpad, nbp, dU, nbp, s));
  PREF_DO(rows_forward_solve(mb, Mpad, dU, dV));
  PREF_DO(launch_row_stats(dV, nbp, M, nbp, dzero, m->se.sf2, dtmp, dvar, s));
  int binfo = 0;
  PREF_HIP(hipMemcpyAsync(&binfo, mb->dinfo, 4, hipMemcpyDeviceToHost, s));
  PREF_HIP(hipMemcpyAsync(mean_out, dmean, (size_t)M * 8, hipMemcpyDeviceToHost, s));
  PREF_HIP(hipMemcpyAsync(var_out, dvar, (size_t)M * 8, hipMemcpyDeviceToHost, s));
  PREF_HIP(hipStreamSynchronize(s));
  if (!rc) rc = info_timeout_rc(binfo);
  if (!rc && binfo > 0) rc = gpk_set_err(-4, "gpk_pref_predict: B lost positive definiteness (minor %d)", binfo);
#undef PREF_HIP
#undef PREF_DO
  dev_free(duvi, (size_t)2 * npairs * 8);
  dev_free(dy, (size_t)npairs * 8);
  dev_free(df, (size_t)np_ * 8);
  dev_free(dc, (size_t)npairs * 8);
  dev_free(dS, (size_t)npairs * nbp * 8);
  dev_free(dZ, (size_t)M * d * 8);
  dev_free(dR, (size_t)Mpad * np_ * 8);
  dev_free(dU, (size_t)Mpad * nbp * 8);
  dev_free(dV, (size_t)Mpad * nbp * 8);
  dev_free(dzero, (size_t)nbp * 8);
  dev_free(dmean, (size_t)Mpad * 8);
  dev_free(dvar, (size_t)Mpad * 8);
  dev_free(dtmp, (size_t)Mpad * 8);
  model_destroy(mb);
  return rc;
}
