/*
 * gpk_bench.h -- the measurement entry points libgpk.so exports NEXT TO its API (include/gpk.h) for bench.py: the timed
 * posterior with inputs resident in HBM, the HIP-event stage timers behind `roofline`, and the two hardware probes whose
 * readings bench.py prints beside it (fp64 MFMA issue peak, shader clock under load).  No integration binds them.
 * Every other probe / kernel-level test hook lives in libgpk_hooks.so (include/gpk_testhooks.h).
 */
#ifndef GPK_BENCH_H
#define GPK_BENCH_H
#include "gpk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement hooks (bench.py) ----------------------------------------- */
/* HIP-event timings (ms) of the stages of the last gpk_fit / gpk_predict on this
 * process; names: "kbuild","potrf","trailing","panel","solve_y","predict".      */
int gpk_last_timing(const char* stage, double* ms_out, double* count_out);
/* algorithmic work (flops for "trailing", bytes for "kbuild") of the launches summed by gpk_last_timing. */
int gpk_last_timing_work(const char* stage, double* work_out);
/* Stage inputs once (X,y,Z resident in HBM) and run fit+predict `reps` times
 * without any host<->device traffic inside the timed region; returns total ms.  */
int gpk_bench_posterior(const double* X, const double* y, int N, int d, const double* log_hyp,
                        const double* Z, int M, int warmup, int reps, double* ms_total_out,
                        double* nlml_out, double* mean_out, double* var_out);

/* probe: shader clock (GHz) sampled on the panel stream while the update stream runs nbig lower updates (m, K = k). */
int gpk_bench_clock_under_load(int m, int k, int nbig, int nsamp, double* ghz_out, double* tflops_out);
/* v_mfma_f64_16x16x4_f64 issue loop with exactly waves_per_simd (1, 2, 4, 8) waves pinned on every SIMD through the LDS
   allocation: the measured fp64 MFMA peak of this device;
   out[6] = TFLOP/s, shader GHz, max / min waves per occupied SIMD, SIMDs occupied, TFLOP/s by the slowest wave's cycles. */
int gpk_bench_mfma_pinned(int waves_per_simd, int iters, double* out);

#ifdef __cplusplus
}
#endif
#endif /* GPK_BENCH_H */
