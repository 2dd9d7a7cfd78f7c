/*
 * gpk_testhooks.h -- test and probe hooks exported by libgpk_hooks.so (gptest_amd/csrc/gpk_hooks.hip), a library of its own
 * on top of libgpk.so that only tests/ and scripts/ load (tests/gpk_hooks.py).  The product library exports none of them.
 *
 * Nothing here replaces a reference interface and no integration binds it: tests/ reach single kernels and the
 * single-process emulation of the block-cyclic job through gpk_test_*, scripts/ probe the hardware through gpk_bench_*.
 * (bench.py's own measurement entry points are in gpk_bench.h and stay in libgpk.so.)
 * Same conventions as gpk.h (host pointers, row-major fp64, 0 = ok).
 */
#ifndef GPK_TESTHOOKS_H
#define GPK_TESTHOOKS_H
#include "gpk.h"
#include "gpk_bench.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- kernel-level test hooks (tests/ only; host buffers, row-major) --------- */
/* C (m x n) = [C -] A (m x k) * B (n x k)^T through the MFMA kernels; tile = 128 (8-wave LDS-DMA kernel) or 64 (the
 * 4-wave kernel: its independent bit-for-bit cross-check).  k1 > 0: the K range split at column k1 over two separately
 * stored operands (the paired trailing update).  xcd_order: force the XCD-aware 8x8-square tile order (tile = 128).
 * tile = 16: the LDS-free one-tile kernel of the panel chain (m = n = k = 128 only; lower: the 16x16 tiles that touch the lower triangle). */
int gpk_test_gemm_nt(const double* A, const double* B, double* C, int m, int n, int k, int sub, int lower, int tile, int k1,
                     int xcd_order);
/* C (m x n) -= A (m x K) B (n x K)^T, K = sum ks[i], with segment i of the K range stored on its own (leading dimension
 * lds[i] >= ks[i]): one launch of the 8-wave kernel over a K-segment table (the deferred corner update). */
int gpk_test_gemm_segs(const double* A, const double* B, double* C, int m, int n, const int* ks, const int* lds, int nseg, int lower);
/* the RCCL calls of the multi-rank path (broadcast on the panel stream, all-gather and all-reduce on the update
 * stream) on a one-rank communicator, payload checked: what a 1-GPU box can verify of gpk_comm.hip. */
int gpk_test_rccl_selfcheck(int n_doubles);
/* single-GPU emulation of the P-rank block-cyclic Cholesky: P virtual rank models driven through the production
 * schedule (potrf_group: per-rank panel / update phases on the panel and update streams, look-ahead), with the
 * RCCL broadcast replaced by device copies from the owner's model.  P = 1: the production one-GPU factorisation. */
int gpk_test_dist_potrf(const double* A, int N, int P, double* L_out);
/* same emulation, every virtual rank builds its own block columns of K(X,X) on the device first. */
int gpk_test_dist_fit(const double* X, int N, int d, const double* log_hyp, double jitter, int P, double* L_out);
/* the whole posterior of a P-rank job on P virtual ranks (fit_group + predict_group, all-gather emulated):
 * nlml_out[P], mean_out / var_out [P][M] = what every rank ends up with. */
int gpk_test_dist_posterior(const double* X, const double* y, int N, int d, const double* log_hyp, const double* Z,
                            int M, int P, double* nlml_out, double* mean_out, double* var_out);
/* probe: us per step of [leaf + nsmall GEMMs of small_m x 128 x 128] on the panel-chain stream, alone / under load. */
int gpk_bench_chain_step(int reps, int nsmall, int small_m, double* us_alone, double* us_contended);
/* device-resident timing of the kernel-matrix builder (mode 0 SE, 1 squared distance, 2 stores only). */
int gpk_bench_se_build(int N, int d, int mode, int lower, int reps, double* ms_out);
/* device-resident timing of the MFMA GEMM: C (m x n) -= A (m x k) B (n x k)^T, `reps` launches; ms per launch. */
int gpk_bench_gemm_nt(int m, int n, int k, int lower, int tile, int reps, double* ms_out);
/* probe: nstreams streams (bit i of high_mask / low_mask: highest / lowest priority, else normal; use_lib_streams: the first three are the
 * library's own update / chain / substitution streams), `reps` small spinning launches (wgs workgroups, us_each microseconds) on each,
 * queued round robin.  out_us[i] = time per launch on stream i, out_us[nstreams] = host wall time per round. */
int gpk_bench_streams(int nstreams, int high_mask, int low_mask, int wgs, int us_each, int reps, int use_lib_streams, double* out_us);
/* streams_abreast (the probe gpk_comm_init runs on a multi-rank process's four streams) on the library's three streams plus n_high
   high-priority and n_normal normal-priority ones; out2 = {worst, best} microseconds per 10 us launch */
int gpk_test_streams_abreast(int n_high, int n_normal, int reps, double* out2);
/* One rank of a P-rank block-cyclic posterior ALONE on the GPU (P virtual ranks in this process; the whole job runs once, then
 * each rank's share is replayed by itself `reps` times, panels of other owners arriving as device copies, never late):
 * ms_out[r] = the best time of rank r, r < P -- a lower bound of the P-GPU job's time, max over r.  comm_stream != 0: those copies
 * on a communication stream of their own, the four-stream layout of a real rank (else they share the substitution stream). */
int gpk_bench_dist_rank_alone(const double* X, const double* y, int N, int d, const double* log_hyp, const double* Z, int M, int P,
                              int reps, int comm_stream, double* ms_out);
/* LDS banking probe for rocprofv3 --pmc (scripts/lds_bank_probe.sh): `reps` ds_read_b64 per lane in one of six address patterns. */
int gpk_bench_lds_bank_probe(int pattern, int reps);
/* leaf kernel timing; dbg_skip bit 0 = skip elimination, bit 1 = skip inverse (timing only), bit 8 = the full-tile form. */
int gpk_bench_leaf(int dbg_skip, int reps, double* us_out);
/* leaf on the chain stream, alone and while the update stream saturates the chip (reps bit 24: the full-tile form,
 * bits 16..23: dbg_skip). */
int gpk_bench_leaf_contended(int reps, double* us_alone, double* us_contended);
/* one chain step of the tile pipeline on a 256 x 256 block B (lower tiles): leaf of tile (0,0), L(1,0) = B(1,0) inv(L(0,0))^T, B(1,1) -= L(1,0) L(1,0)^T.
 * mode 0: three launches; 1: the fused kernel; 2: the fused kernel polling a flag that the update stream sets meanwhile; 3: a flag nobody sets
 * (returns -3 once the bounded wait gives up).  L_out 256 x 128, Dinv_out 128 x 128, A11_out 128 x 128.  Returns info. */
int gpk_test_chain_step(const double* B, double* L_out, double* Dinv_out, double* A11_out, int mode);
/* factor one 128x128 SPD tile: L_out (128x128 lower) and Dinv_out = inv(L); full_tile: the 1024-thread form instead of
 * the slot-sized one (bit-identical results).  Returns info. */
int gpk_test_potf2_128(const double* A, double* L_out, double* Dinv_out, int full_tile);

#ifdef __cplusplus
}
#endif
#endif /* GPK_TESTHOOKS_H */
