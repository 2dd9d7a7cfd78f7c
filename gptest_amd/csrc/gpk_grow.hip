// Incremental data growth: the reference's GP_parameter_fit.py:62 feeds its model one more
// observation per frame, gpm.set_XY(X[0:ii+1], Y[0:ii+1]), and GPy refactors all of K every time.  gpk_append extends
// the Cholesky factor held in HBM by k new points: with t0 the first row of the 128-tile that contains the first new
// row, only the tile rows >= t0 are recomputed,
//     V = K[t0:, 0:t0] L11^-T          the row substitution of the prediction path on the MFMA GEMM (K = 128 steps)
//     S = K[t0:, t0:] - V V^T          ONE lower SYRK with K = t0
//     L22 = chol(S)                    the ordinary panel factorisation on the (<= 128 + k)-wide trailing block
// i.e. O((128 + k) n^2) flops instead of n^3/3, with the factor of the first t0 columns left where it is.
// Needs a model fitted with flag 4 ("growable": uniform 128-wide panels, a layout that does not depend on N).
#include "../../include/gpk.h"
#include "../../include/gpk_bench.h"
#include "gpk_core.h"
#include <cmath>

extern "C" int gpk_append(gpk_model** handle_inout, const double* X_new, const double* y_new, int k, double* nlml_out) {
  GPK_TRY(rt_require());
  if (!handle_inout || !*handle_inout || !X_new || !y_new) return gpk_set_err(-1, "gpk_append: null pointer");
  gpk_model* mo = *handle_inout;
  if (k < 1) return gpk_set_err(-1, "gpk_append: k=%d must be >= 1", k);
  if (!mo->growable || mo->P != 1) return gpk_set_err(-1, "gpk_append: fit the model with flag 4 (growable, single GPU)");
  if (!mo->factored || !mo->has_y) return gpk_set_err(-1, "gpk_append: the model holds no fitted posterior");
  hipStream_t s = G.s_main;
  const int d = mo->d, n_old = mo->N, n_new = n_old + k;
  const int t0 = n_old / 128 * 128, p0 = t0 / 128;
  // ---- a model of the new size; when the padded size does not change the old one is extended in place
  gpk_model* m = mo;
  const bool realloc_model = round_up(n_new, GPK_T) != mo->Npad;
  if (realloc_model) {
    GPK_TRY(model_create(&m, n_new, d, false, -1, -1, true));
    m->se = mo->se; m->sn2 = mo->sn2; m->jitter = mo->jitter; m->has_y = true;
  }
  struct Guard {  // on failure the caller keeps the old, untouched handle (in-place failures invalidate it)
    gpk_model *fresh, *old;
    bool ok = false;
    ~Guard() {
      if (!ok && fresh != old) { (void)hipStreamSynchronize(G.s_main); model_destroy(fresh); }
    }
  } guard{m, mo};
  const int np_ = m->Npad;
  auto body = [&]() -> int {
    if (realloc_model) {
      GPK_HIP(hipMemcpyAsync(m->dX, mo->dX, (size_t)n_old * d * 8, hipMemcpyDeviceToDevice, s));
      GPK_HIP(hipMemsetAsync(m->dy, 0, (size_t)np_ * 8, s));
      GPK_HIP(hipMemcpyAsync(m->dy, mo->dy, (size_t)n_old * 8, hipMemcpyDeviceToDevice, s));
      // columns < t0 of the factor: panel p is (Npad - c0) x 128 row-major, so its old rows are one contiguous block
      for (int p = 0; p < p0; ++p)
        GPK_HIP(hipMemcpyAsync(m->Lpanel(p), mo->Lpanel(p), (size_t)(mo->Npad - mo->pc0[p]) * 128 * 8, hipMemcpyDeviceToDevice, s));
      GPK_HIP(hipMemcpyAsync(m->dDinv, mo->dDinv, (size_t)p0 * 128 * 128 * 8, hipMemcpyDeviceToDevice, s));
    }
    GPK_HIP(hipMemcpyAsync(m->dX + (size_t)n_old * d, X_new, (size_t)k * d * 8, hipMemcpyHostToDevice, s));
    GPK_HIP(hipMemcpyAsync(m->dy + n_old, y_new, (size_t)k * 8, hipMemcpyHostToDevice, s));
    m->N = n_new;
    const int rows = np_ - t0;  // tile rows to (re)compute, a multiple of 128
    // K[t0:, t0:] (lower tiles; identity on the padding) straight into the trailing block of dK
    GPK_TRY(launch_se_build(m->dX + (size_t)t0 * d, n_new - t0, rows, m->dX + (size_t)t0 * d, n_new - t0, rows, m->se, 1, 0,
                            m->dK + (size_t)t0 * m->ldk + t0, m->ldk, 1, s, 0, m->dX));
    if (p0 > 0) {
      Scratch sc;
      double *dR = nullptr, *dV = nullptr;
      GPK_TRY(sc.get(&dR, (size_t)rows * np_));
      GPK_TRY(sc.get(&dV, (size_t)rows * np_));
      // R = K(X[t0:], X[0:t0]) in the first t0 columns of a rows x Npad row block
      GPK_TRY(launch_se_build(m->dX + (size_t)t0 * d, n_new - t0, rows, m->dX, t0, t0, m->se, 0, 0, dR, np_, 0, s));
      // V = R L11^-T: the row substitution over the untouched panels, columns < t0 only
      for (int p = 0; p < p0; ++p) {
        const int c0 = p * 128;
        GemmNT g{};
        g.A = dR + c0; g.lda = np_;
        g.B = m->dDinv + (size_t)p * 128 * 128; g.ldb = 128;
        g.C = dV + c0; g.ldc = np_;
        g.m = rows; g.n = 128; g.k = 128; g.sub = 0;
        GPK_TRY(launch_gemm_nt(g, false, s));
        if (t0 - c0 - 128 > 0) {  // R[:, c0+128 : t0] -= V[:, panel p] L[c0+128 : t0, panel p]^T
          GemmNT u{};
          u.A = dV + c0; u.lda = np_;
          u.B = m->Lpanel(p) + (size_t)128 * 128; u.ldb = 128;
          u.C = dR + c0 + 128; u.ldc = np_;
          u.m = rows; u.n = t0 - c0 - 128; u.k = 128; u.sub = 1;
          GPK_TRY(launch_gemm_nt(u, false, s));
        }
        // rows t0.. of panel p are V[:, panel p]
        GPK_HIP(hipMemcpy2DAsync(m->Lpanel(p) + (size_t)(t0 - c0) * 128, 128 * 8, dV + c0, (size_t)np_ * 8, 128 * 8, rows,
                                 hipMemcpyDeviceToDevice, s));
      }
      // S = K[t0:, t0:] - V V^T, one lower update with K = t0
      GemmNT g{};
      g.A = dV; g.lda = np_; g.B = dV; g.ldb = np_;
      g.C = m->dK + (size_t)t0 * m->ldk + t0; g.ldc = m->ldk;
      g.m = rows; g.n = rows; g.k = t0; g.sub = 1; g.lower = 1;
      GPK_TRY(launch_gemm_nt(g, false, s));
      GPK_HIP(hipStreamSynchronize(s));  // dR / dV go back to the pool
    }
    // factor the trailing block: the ordinary right-looking panel loop from panel p0 on
    GPK_HIP(hipMemsetAsync(m->dinfo, 0, 16, s));
    for (int p = p0; p < m->nP; ++p) {
      GPK_TRY(factor_panel(m, p, s));
      const int r0 = m->pc0[p] + m->pw[p];
      GPK_TRY(panel_update(m, p, r0, r0, np_ - r0, 1, s));
    }
    // scalars: log det, w = L^-1 y (O(n^2)), w.w
    GPK_TRY(launch_sum_log_diag(m->dDinv, n_new, m->dscal + 0, s));
    GPK_HIP(hipMemcpyAsync(m->dw, m->dy, (size_t)np_ * 8, hipMemcpyDeviceToDevice, s));
    GPK_TRY(trsv_forward(m, m->dw));
    GPK_TRY(launch_dot(m->dw, m->dw, np_, m->dscal + 1, s));
    double hs[2] = {0, 0};
    int hinfo = 0;
    GPK_HIP(hipMemcpyAsync(hs, m->dscal, 16, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipMemcpyAsync(&hinfo, m->dinfo, 4, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));
    m->info = hinfo;
    GPK_TRY(info_timeout_rc(hinfo));
    if (hinfo > 0) {
      m->factored = false;
      return gpk_set_err(hinfo > n_new ? n_new : hinfo, "kernel matrix is not positive definite after the append (leading minor %d)", hinfo);
    }
    m->factored = true;
    m->logdet = hs[0];
    m->quad = hs[1];
    m->nlml = 0.5 * m->quad + m->logdet + 0.5 * n_new * std::log(2.0 * M_PI);
    return 0;
  };
  int rc = body();
  if (rc) {
    if (!realloc_model) { mo->N = n_old; mo->factored = false; }  // extended in place: the handle is no longer usable
    return rc;
  }
  guard.ok = true;
  if (realloc_model) {
    GPK_HIP(hipStreamSynchronize(s));
    model_destroy(mo);
    *handle_inout = m;
  }
  if (nlml_out) *nlml_out = m->nlml;
  return 0;
}
