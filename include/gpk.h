/*
 * gpk.h -- C ABI of the MI355X-native GP posterior core (libgpk.so).
 *
 * The reference (osurdml/GPtest) has no FFI: its boundary is the Python module
 * surface of GPr.py / GPpref.py.  This header is the boundary a replacement
 * shared library must export; gptest_amd/GPr.py and gptest_amd/GPpref.py bind
 * it with ctypes.  Every entry point cites the reference
 * interface it replaces.
 *
 * Conventions
 *   - all pointers are caller-owned HOST memory, C-contiguous (row-major) fp64
 *     unless noted; inputs are never modified unless documented;
 *   - return 0 = ok; < 0 = API / HIP / RCCL error (text via gpk_last_error());
 *     > 0 = LAPACK-style "leading minor of order k is not positive definite"
 *     (the Python side raises numpy.linalg.LinAlgError, which is the only
 *     error the reference handles: GPpref.py:126-135);
 *   - blocking, one HIP device per process, handles are not thread-safe
 *     (the reference is single-threaded);
 *   - log_hyp layout = log([l_1..l_d, sigma_f, sigma_n])  (GPr.py:90-97).
 *     A log sigma_n of -inf gives sn2 = 0 (GPpref's noise-free K, GPpref.py:122).
 */
#ifndef GPK_H
#define GPK_H

#ifdef __cplusplus
extern "C" {
#endif

#define GPK_ABI_VERSION 1
#define GPK_UNIQUE_ID_BYTES 128

/* ---- runtime ------------------------------------------------------------ */
int         gpk_abi_version(void);
int         gpk_device_count(void);          /* no HIP context is created   */
int         gpk_init(int device);            /* bind this process to a GPU  */
void        gpk_shutdown(void);
const char* gpk_last_error(void);
int         gpk_sync(void);                  /* hipDeviceSynchronize        */
int         gpk_set_option(const char* key, double value); /* schedule switches: "nb", "lookahead", "tail_rows",
                                                               "pred_defer_from", "corner_from", "chain_fused", ...
                                                               (README.md lists them); results do not depend on them
                                                               beyond rounding, most not at all */
double      gpk_get_option(const char* key);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ---------------------- */
/* rank 0 calls gpk_comm_unique_id, the host broadcasts the 128 bytes by any
 * means (gptest_amd/dist.py: a plain TCP socket next to MASTER_PORT), every rank
 * calls gpk_comm_init.  A multi-rank process gets a communication stream: the
 * panel broadcasts run there, off the panel chain; y and this rank's shard of
 * the prediction are substituted on the substitution stream behind the
 * per-panel "available" events (option "dist_pipeline", default 1; 0: three
 * streams, prediction after the factorisation).  gpk_comm_destroy gives the
 * process its single-GPU stream layout back.
 * Option "dist_group": panels per ownership block (default 2).              */
int gpk_comm_unique_id(void* out128);
int gpk_comm_init(int rank, int nranks, const void* uid128);
int gpk_comm_rank(void);
int gpk_comm_size(void);
int gpk_comm_barrier(void);
int gpk_comm_allreduce_max(double* inout, int n);   /* host doubles, via device */
int gpk_comm_destroy(void);

/* ---- element-wise builders ------------------------------------------------ */
/* squared_distance(A,B): GPr.py:4-13 (dup GPpref.py:13-22), extended to (n,d). */
int gpk_sqdist(const double* A, int n, const double* B, int m, int d, double* out /* n*m */);
/* SquaredExponential.compute_Kxx_matrix: GPr.py:99-103.  K = (sn2+jitter) I + sf2 exp(-D/2). */
int gpk_se_kxx(const double* X, int N, int d, const double* log_hyp, double jitter, double* K_out /* N*N */);
/* SquaredExponential.compute_Kxz_matrix: GPr.py:105-110. */
int gpk_se_kxz(const double* X, int N, const double* Z, int M, int d, const double* log_hyp, double* Kxz_out /* N*M */);

/* ---- factorisation -------------------------------------------------------- */
/* np.linalg.cholesky(A): GPr.py:62, GPpref.py:128.  In: symmetric (lower read);
 * out: lower-triangular L, strict upper zeroed.                               */
int gpk_potrf(double* A, int N);

/* ---- fused model entry points --------------------------------------------- */
typedef struct gpk_model gpk_model;

/* GaussianProcess.compute_likelihood: GPr.py:57-69.  Builds K on the device,
 * factors it, w = L^-1 y, nlml = w.w/2 + sum log L_ii + N log(2 pi)/2.  K and L
 * stay in HBM behind the handle.  y may be NULL (factor only; nlml_out gets
 * sum log L_ii, the reference's "logdetK" of GPpref.py:131).
 * flags: bit0 = keep an unfactored copy of K in HBM (needed by gpk_kmatvec / gpk_kgather);
 *        bit1 (with bit0, y = NULL) = build and keep K only, factor nothing (gpk_class_laplace factors B instead);
 *        bit2 = growable (gpk_append): uniform 128-wide panels instead of the automatic panel schedule. */
int gpk_fit(const double* X, const double* y, int N, int d, const double* log_hyp,
            double jitter, int flags, gpk_model** handle_out, double* nlml_out);
/* GaussianProcess.compute_prediction: GPr.py:45-54.  mean = Kzx K^-1 y,
 * var = sf2 - diag(Kzx K^-1 Kxz)  (latent variance, no +sn2: GPr.py:51).      */
int gpk_predict(gpk_model* h, const double* Z, int M, double* mean_out, double* var_out);
/* compute_likelihood + compute_prediction in one call (GPr.py:45-69) sharing one K build and one Cholesky;
 * the forward substitution of the prediction is pipelined panel by panel behind the factorisation. */
int gpk_fit_predict(const double* X, const double* y, int N, int d, const double* log_hyp, double jitter,
                    const double* Z, int M, gpk_model** handle_out, double* nlml_out, double* mean_out,
                    double* var_out);
/* K^-1 B for B (N, nrhs) row-major: replaces the explicit iK of GPr.py:63 / GPpref.py:129. */
int gpk_solve(gpk_model* h, const double* B, int nrhs, double* X_out);
/* L^-1 B (forward substitution only), B (N, nrhs) row-major. */
int gpk_solve_lower(gpk_model* h, const double* B, int nrhs, double* X_out);
/* K B with the kept copy of K (jittered/noisy diagonal included), B (N, nrhs). */
int gpk_kmatvec(gpk_model* h, const double* B, int nrhs, double* out);
/* out[p*nidx+q] = K[idx[p], idx[q]] from the kept copy (GPpref Newton step, GPpref.py:142-143). */
int gpk_kgather(gpk_model* h, const long long* idx, int nidx, double* out);
/* Gradient of the nlml of GPr.py:57-69 with respect to the d+2 log hyper-parameters (same order as log_hyp:
 * d length scales, signal sd, noise sd), re-using the factor of the fit.  The reference has no gradient (its
 * demos hand-set hyp); this is the "next" row of SURVEY 8(f) for fitting hyper-parameters.  Single-GPU models,
 * d <= 32 like the kernel-matrix builder. */
int gpk_nlml_grad(gpk_model* h, double* grad_out);
/* scalars of the last fit: which = 0 nlml, 1 sum log L_ii, 2 y^T K^-1 y, 3 sf2, 4 sn2. */
int gpk_model_scalar(gpk_model* h, int which, double* out);
/* copy L out as dense lower-triangular (N,N) row-major. */
int gpk_model_get_L(gpk_model* h, double* L_out);
int gpk_free(gpk_model* h);

/* PreferenceGaussianProcess.calc_laplace, Newton loop + PrefProbit.derivatives/log_marginal
 * (GPpref.py:137-157, 56-94) on a model fitted with flags bit0 (K + eps*I factored, copy kept).
 * uvi: (npairs,2) int64 row-major indices into the training points, y: +-1 labels, sigma: probit noise
 * (the reference always ends up with 1.0, GPpref.py:115), f_inout: (nx) start / mode, ferr_trace: optional
 * max_iter doubles (max|f_new - f| per iteration).  Stops when max|df| <= delta_f or after max_iter. */
int gpk_pref_laplace(gpk_model* h, const long long* uvi, const double* y, int npairs, double sigma,
                     double* f_inout, double delta_f, int max_iter, double* lml_out, int* iters_out,
                     double* ferr_trace);
/* PrefProbit's per-pair terms (GPpref.py:56-58, 68-80, 90-92) for callers of the likelihood object itself: z_p, the
 * gradient weight q_p (dpy_df gets -q_p at u_p and +q_p at v_p), the curvature c_p (W gets +c_p at (u,u), (v,v) and -c_p at
 * (u,v), (v,u): c = -inner of GPpref.py:80) and log Phi(z_p), each (npairs) or NULL.  f: (nx).  No model handle needed. */
int gpk_pref_pair_stats(const long long* uvi, const double* y, int npairs, double sigma, const double* f, int nx,
                        double* z_out, double* q_out, double* c_out, double* logphi_out);
/* Latent posterior at M test inputs Z (M, d) under the Laplace approximation at f_hat (the mode gpk_pref_laplace
 * returned): mean = k*^T (K+eps I)^-1 f_hat, var = k** - k*^T (K + eps I + W^-1)^-1 k*.  The reference stops at
 * the mode (GPpref.py:157) and never predicts; SURVEY 8(f) item 2 names this as the next step.  Same handle and
 * pair arguments as gpk_pref_laplace. */
int gpk_pref_predict(gpk_model* h, const long long* uvi, const double* y, int npairs, double sigma,
                     const double* f_hat, const double* Z, int M, double* mean_out, double* var_out);

/* ---- incremental data growth ---------------------------------------------------------- */
/* GP_parameter_fit.py:62 grows its model one observation per frame (gpm.set_XY(X[0:ii+1], Y[0:ii+1])) and GPy
 * refactors K every time.  gpk_append extends the factor of a model fitted with gpk_fit flag bit2 (4, "growable":
 * uniform 128-wide panels) by k more points: row substitution of the new tile rows + one SYRK + the panel
 * factorisation of the trailing (<= 128 + k)-wide block, O((128 + k) n^2) instead of n^3/3.  The handle may be
 * replaced (a larger padded size); on a positive return (not positive definite) the OLD handle stays valid only
 * if it was replaced, otherwise it must be freed.  nlml_out: the nlml of the grown model. */
int gpk_append(gpk_model** handle_inout, const double* X_new, const double* y_new, int k, double* nlml_out);

/* ---- binary GP classification, Laplace approximation ----------------------------------- */
/* What GPc.py:40-55 (a non-parsing sketch citing Rasmussen eq. 3.12) and GP_classification_demo.py:5-50 head to:
 * Rasmussen & Williams Alg. 3.1 on a handle from gpk_fit(..., flags = 3) (K built and kept, nothing factored).
 * y in {-1,+1} (GPc.py:24-38 maps {0,1} to it), link 0 = probit (NormCDF, GPc.py:4-6), 1 = logit (GPc.py:13-14).
 * f_inout: start / mode (N).  Stops when max|f_new - f| <= delta_f (the rule of GPpref.py:140) or after max_iter.
 * lml_out = -a^T f/2 + sum log p(y|f) - sum log diag L_B (R&W eq. 3.32).  PARITY UNPINNED (no runnable reference). */
int gpk_class_laplace(gpk_model* h, const double* y, int link, double* f_inout, double delta_f, int max_iter,
                      double* lml_out, int* iters_out, double* ferr_trace /* max_iter or NULL */);
/* R&W Alg. 3.2 at the mode: latent mean / variance and the averaged probability of class +1 at M test points. */
int gpk_class_predict(gpk_model* h, const double* y, int link, const double* f_hat, const double* Z, int M,
                      double* mean_out, double* var_out, double* prob_out /* or NULL */);

#ifdef __cplusplus
}
#endif
#endif /* GPK_H */
