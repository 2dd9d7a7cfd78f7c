// Host driver of binary GP classification under the Laplace approximation (the loop the
// reference's GPc.py:40-55 sketches with "Rasmussen eq. 3.12" and GP_classification_demo.py:5-50 sets the data up for).
// Rasmussen & Williams Alg. 3.1 (mode) and 3.2 (prediction) on a K-only model handle (gpk_fit flags = 3: K built and
// kept in HBM, nothing factored): per Newton iteration ONE Cholesky, of B = I + W^1/2 K W^1/2 (n x n, eigenvalues >= 1),
// on the same MFMA factorisation as regression; K itself is only ever multiplied.  Everything stays in HBM; per
// iteration the host sees one scalar (max |f_new - f|, the stopping rule of the sibling loop GPpref.py:140).
#include "../../include/gpk.h"
#include "../../include/gpk_bench.h"
#include "gpk_core.h"
#include <cmath>

struct ClassState {
  gpk_model* mb = nullptr;  // B and its factor
  double *dy = nullptr, *df = nullptr, *dfn = nullptr, *dg = nullptr, *dw = nullptr, *dsw = nullptr, *dlp = nullptr,
         *db = nullptr, *dkb = nullptr, *dr = nullptr, *da = nullptr, *dscal = nullptr;
  ~ClassState() {
    if (mb) model_destroy(mb);
  }
};

static int class_alloc(Scratch& sc, ClassState& st, gpk_model* m) {
  const size_t np_ = m->Npad;
  GPK_TRY(model_create(&st.mb, m->N, 1, false, 1, 0));
  if (st.mb->Npad != m->Npad) return gpk_set_err(-1, "internal: padded sizes of K and B differ");
  GPK_TRY(sc.get(&st.dy, np_));
  GPK_TRY(sc.get(&st.df, np_));
  GPK_TRY(sc.get(&st.dfn, np_));
  GPK_TRY(sc.get(&st.dg, np_));
  GPK_TRY(sc.get(&st.dw, np_));
  GPK_TRY(sc.get(&st.dsw, np_));
  GPK_TRY(sc.get(&st.dlp, np_));
  GPK_TRY(sc.get(&st.db, np_));
  GPK_TRY(sc.get(&st.dkb, np_));
  GPK_TRY(sc.get(&st.dr, np_));
  GPK_TRY(sc.get(&st.da, np_));
  GPK_TRY(sc.get(&st.dscal, (size_t)8));
  return 0;
}

// labels to the device (+-1 checked on the host), latent vector zero-padded
static int class_stage(ClassState& st, gpk_model* m, const double* y, const double* f, hipStream_t s) {
  const size_t np_ = m->Npad, n = m->N;
  for (size_t i = 0; i < n; ++i)
    if (y[i] != 1.0 && y[i] != -1.0) return gpk_set_err(-1, "labels must be -1 or +1 (y[%zu] = %g)", i, y[i]);
  double* all[] = {st.dy, st.df, st.dfn, st.dg, st.dw, st.dsw, st.dlp, st.db, st.dkb, st.dr, st.da};
  for (double* p : all) GPK_HIP(hipMemsetAsync(p, 0, np_ * 8, s));
  GPK_HIP(hipMemcpyAsync(st.dy, y, n * 8, hipMemcpyHostToDevice, s));
  GPK_HIP(hipMemcpyAsync(st.df, f, n * 8, hipMemcpyHostToDevice, s));
  return 0;
}

// statistics at st.df and B = I + W^1/2 K W^1/2 factored in st.mb; r <- L_B^-1 r rides on the factorisation if given
static int class_factor_B(ClassState& st, gpk_model* m, int link, double* r_forward, hipStream_t s) {
  const int n = m->N, np_ = m->Npad;
  GPK_TRY(launch_class_stats(st.dy, st.df, n, link, st.dg, st.dw, st.dsw, st.dlp, s));
  GPK_TRY(launch_pref_build_B(m->dKcopy, np_, st.dw, n, np_, st.mb->dK, st.mb->ldk, s));
  if (r_forward) return potrf_forward(st.mb, r_forward);
  return potrf_device(st.mb);
}

extern "C" int gpk_class_laplace(gpk_model* m, const double* y, int link, double* f_inout, double delta_f, int max_iter,
                                 double* lml_out, int* iters_out, double* ferr_trace) {
  GPK_TRY(rt_require());
  if (!m || !y || !f_inout) return gpk_set_err(-1, "gpk_class_laplace: null pointer");
  if (!m->dKcopy) return gpk_set_err(-1, "gpk_class_laplace: fit the model with the keep-K flag (gpk_fit flags 1 or 3)");
  if (m->P != 1) return gpk_set_err(-1, "gpk_class_laplace: single-GPU models only");
  if (max_iter < 1 || (link != 0 && link != 1)) return gpk_set_err(-1, "gpk_class_laplace: bad arguments");
  hipStream_t s = G.s_main;
  const int n = m->N, np_ = m->Npad;
  ClassState st;
  Scratch sc;
  GPK_TRY(class_alloc(sc, st, m));
  GPK_TRY(class_stage(st, m, y, f_inout, s));
  int iters = 0;
  for (int it = 0; it < max_iter; ++it) {
    // R&W Alg. 3.1 lines 4-9:  b = W f + grad;  a = b - W^1/2 B^-1 W^1/2 K b;  f <- K a
    GPK_TRY(launch_class_stats(st.dy, st.df, n, link, st.dg, st.dw, st.dsw, nullptr, s));
    GPK_TRY(launch_class_ew(st.dw, st.df, st.dg, n, 0, st.db, s));
    GPK_TRY(launch_gemv(m->dKcopy, np_, np_, np_, st.db, st.dkb, 1.0, 0.0, s));
    GPK_TRY(launch_class_ew(st.dsw, st.dkb, nullptr, n, 2, st.dr, s));
    GPK_TRY(launch_pref_build_B(m->dKcopy, np_, st.dw, n, np_, st.mb->dK, st.mb->ldk, s));
    GPK_TRY(potrf_forward(st.mb, st.dr));
    GPK_TRY(trsv_backward(st.mb, st.dr));
    GPK_TRY(launch_class_ew(st.dsw, st.dr, st.db, n, 1, st.da, s));
    GPK_TRY(launch_gemv(m->dKcopy, np_, np_, np_, st.da, st.dfn, 1.0, 0.0, s));
    GPK_TRY(launch_maxabsdiff(st.dfn, st.df, n, st.dscal, s));
    double ferr = 0.0;
    int binfo = 0;
    GPK_HIP(hipMemcpyAsync(&ferr, st.dscal, 8, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipMemcpyAsync(&binfo, st.mb->dinfo, 4, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));
    GPK_TRY(info_timeout_rc(binfo));
    if (binfo > 0) return gpk_set_err(-4, "Newton system lost positive definiteness (minor %d)", binfo);
    std::swap(st.df, st.dfn);
    iters = it + 1;
    if (ferr_trace) ferr_trace[it] = ferr;
    if (!(ferr > delta_f)) break;
  }
  // R&W Alg. 3.1 line 10 / eq. 3.32:  log q(y|X) = -a^T f / 2 + sum log p(y_i|f_i) - sum log diag L_B, with a and L_B of
  // the last iteration and f = K a
  double hs[3] = {0, 0, 0};
  GPK_TRY(launch_class_stats(st.dy, st.df, n, link, nullptr, nullptr, nullptr, st.dlp, s));
  GPK_TRY(launch_sum(st.dlp, n, st.dscal, s));
  GPK_TRY(launch_dot(st.da, st.df, np_, st.dscal + 1, s));
  GPK_TRY(launch_sum_log_diag(st.mb->dDinv, n, st.dscal + 2, s));
  GPK_HIP(hipMemcpyAsync(hs, st.dscal, 24, hipMemcpyDeviceToHost, s));
  GPK_HIP(hipMemcpyAsync(f_inout, st.df, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  GPK_HIP(hipStreamSynchronize(s));
  if (lml_out) *lml_out = hs[0] - 0.5 * hs[1] - hs[2];
  if (iters_out) *iters_out = iters;
  return 0;
}

// R&W Alg. 3.2 at the mode f_hat:  mean* = k*^T grad log p(y|f_hat),  var* = k** - |L_B^-1 W^1/2 k*|^2,
// prob* = the averaged predictive probability of class +1 (class_prob_kernel).
extern "C" int gpk_class_predict(gpk_model* m, const double* y, int link, const double* f_hat, const double* Z, int M,
                                 double* mean_out, double* var_out, double* prob_out) {
  GPK_TRY(rt_require());
  if (!m || !y || !f_hat || !Z || !mean_out || !var_out) return gpk_set_err(-1, "gpk_class_predict: null pointer");
  if (!m->dKcopy) return gpk_set_err(-1, "gpk_class_predict: fit the model with the keep-K flag (gpk_fit flags 1 or 3)");
  if (m->P != 1) return gpk_set_err(-1, "gpk_class_predict: single-GPU models only");
  if (M < 1 || (link != 0 && link != 1)) return gpk_set_err(-1, "gpk_class_predict: bad arguments");
  hipStream_t s = G.s_main;
  const int n = m->N, np_ = m->Npad, d = m->d, Mpad = round_up(M, 128);
  ClassState st;
  Scratch sc;
  GPK_TRY(class_alloc(sc, st, m));
  GPK_TRY(class_stage(st, m, y, f_hat, s));
  double *dZ = nullptr, *dR = nullptr, *dU = nullptr, *dV = nullptr, *dmean = nullptr, *dvar = nullptr, *dtmp = nullptr,
         *dprob = nullptr;
  GPK_TRY(sc.get(&dZ, (size_t)M * d));
  GPK_TRY(sc.get(&dR, (size_t)Mpad * np_));
  GPK_TRY(sc.get(&dU, (size_t)Mpad * np_));
  GPK_TRY(sc.get(&dV, (size_t)Mpad * np_));
  GPK_TRY(sc.get(&dmean, (size_t)Mpad));
  GPK_TRY(sc.get(&dvar, (size_t)Mpad));
  GPK_TRY(sc.get(&dtmp, (size_t)Mpad));
  GPK_TRY(sc.get(&dprob, (size_t)Mpad));
  GPK_HIP(hipMemcpyAsync(dZ, Z, (size_t)M * d * 8, hipMemcpyHostToDevice, s));
  GPK_TRY(class_factor_B(st, m, link, nullptr, s));
  GPK_TRY(launch_se_build(dZ, M, Mpad, m->dX, n, np_, m->se, 0, 0, dR, np_, 0, s));
  GPK_TRY(launch_gemv(dR, np_, M, np_, st.dg, dmean, 1.0, 0.0, s));
  GPK_TRY(launch_class_scale_rows(dR, np_, st.dsw, n, M, Mpad, np_, dU, np_, s));
  GPK_TRY(rows_forward_solve(st.mb, Mpad, dU, dV));
  GPK_TRY(launch_row_stats(dV, np_, M, np_, st.dr /* zeros */, m->se.sf2, dtmp, dvar, s));
  GPK_TRY(launch_class_prob(dmean, dvar, M, link, dprob, s));
  int binfo = 0;
  GPK_HIP(hipMemcpyAsync(&binfo, st.mb->dinfo, 4, hipMemcpyDeviceToHost, s));
  GPK_HIP(hipMemcpyAsync(mean_out, dmean, (size_t)M * 8, hipMemcpyDeviceToHost, s));
  GPK_HIP(hipMemcpyAsync(var_out, dvar, (size_t)M * 8, hipMemcpyDeviceToHost, s));
  if (prob_out) GPK_HIP(hipMemcpyAsync(prob_out, dprob, (size_t)M * 8, hipMemcpyDeviceToHost, s));
  GPK_HIP(hipStreamSynchronize(s));
  GPK_TRY(info_timeout_rc(binfo));
  if (binfo > 0) return gpk_set_err(-4, "gpk_class_predict: B lost positive definiteness (minor %d)", binfo);
  return 0;
}
