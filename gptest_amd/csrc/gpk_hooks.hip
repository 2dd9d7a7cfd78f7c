// Measurement and test hooks (include/gpk_testhooks.h) -> libgpk_hooks.so, which links libgpk.so and is loaded by tests/ and
// scripts/ only: single kernels behind host buffers, the single-process emulation of the block-cyclic job, hardware probes.
// Nothing here is part of the product library (nm -D libgpk.so shows no gpk_test_* symbol) and no integration binds it.
#include "../../include/gpk.h"
#include "../../include/gpk_testhooks.h"
#include "gpk_core.h"
#include <cmath>
#include <map>
#include <chrono>
#include <rccl/rccl.h>

const char* gpk_err_cstr();

// ---------------------------------------------------------------- kernel-level test hooks
extern "C" {

// Single-GPU emulation of a P-rank block-cyclic job: P rank models live in this process and are driven through
// the production group entry points (potrf_group / fit_group / predict_group), i.e. the very phases, stream and
// event order a real rank runs; only the RCCL broadcast / all-gather are device copies from the peer's model.
struct VirtualJob {
  std::vector<gpk_model*> ms;
  ~VirtualJob() {
    for (gpk_model* m : ms)
      if (m) model_destroy(m);
  }
  int create(int N, int d, int P) {
    ms.assign(P, nullptr);
    for (int r = 0; r < P; ++r) GPK_TRY(model_create(&ms[r], N, d, false, P, r));
    for (int r = 0; r < P; ++r) ms[r]->vgroup = ms.data();
    return 0;
  }
};

static int virtual_job_check_info(VirtualJob& job, int N) {
  hipStream_t s = G.s_main;
  int worst = 0;
  for (gpk_model* m : job.ms) {
    int hinfo = 0;
    GPK_HIP(hipMemcpyAsync(&hinfo, m->dinfo, 4, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));
    GPK_TRY(info_timeout_rc(hinfo));
    if (hinfo > worst) worst = hinfo;
  }
  if (worst > 0) return worst > N ? N : worst;
  return 0;
}

// L of rank P-1 is returned.  P = 1 is the production single-GPU factorisation (potrf_device).
int gpk_test_dist_potrf(const double* A, int N, int P, double* L_out) {
  GPK_TRY(rt_require());
  if (!A || !L_out || N < 1 || P < 1 || P > 16) return gpk_set_err(-1, "gpk_test_dist_potrf: bad arguments");
  hipStream_t s = G.s_main;
  VirtualJob job;
  GPK_TRY(job.create(N, 1, P));
  const int np_ = job.ms[0]->Npad;
  for (int r = 0; r < P; ++r) {
    gpk_model* m = job.ms[r];
    GPK_HIP(hipMemsetAsync(m->dK, 0, (size_t)np_ * m->ldk * 8, s));
    if (P == 1) {
      GPK_HIP(hipMemcpy2DAsync(m->dK, (size_t)m->ldk * 8, A, (size_t)N * 8, (size_t)N * 8, N, hipMemcpyHostToDevice, s));
      if (np_ > N) GPK_TRY(launch_pad_identity(m->dK, m->ldk, N, np_, s));
      continue;
    }
    for (const gpk_model::ColRange& cr : m->owned_ranges()) {  // the rank's block columns of A (cyclic blocks, then the tail block on its owner)
      const int nc = std::min(cr.ncols, N - cr.gc0);
      if (nc > 0)
        GPK_HIP(hipMemcpy2DAsync(m->dK + cr.lc0, (size_t)m->ldk * 8, A + cr.gc0, (size_t)N * 8, (size_t)nc * 8, N, hipMemcpyHostToDevice, s));
      const int g0 = std::max(N, cr.gc0), cnt = cr.gc0 + cr.ncols - g0;  // ones on the padded diagonal entries of this block
      if (cnt > 0) GPK_TRY(launch_set_diag(m->dK + (size_t)g0 * m->ldk + cr.lc0 + (g0 - cr.gc0), m->ldk, cnt, 1.0, s));
    }
  }
  GPK_TRY(P == 1 ? potrf_device(job.ms[0]) : potrf_group(job.ms.data(), P));
  GPK_TRY(virtual_job_check_info(job, N));
  return model_unpack_L(job.ms[P - 1], L_out);
}

int gpk_test_gemm_nt(const double* A, const double* B, double* C, int m, int n, int k, int sub, int lower, int tile, int k1,
                     int xcd_order) {
  GPK_TRY(rt_require());
  if (!A || !B || !C || m < 1 || n < 1 || k < 1) return gpk_set_err(-1, "gpk_test_gemm_nt: bad arguments");
  hipStream_t s = G.s_main;
  Scratch sc;
  double *dA = nullptr, *dB = nullptr, *dC = nullptr, *dA2 = nullptr, *dB2 = nullptr;
  GPK_TRY(sc.get(&dA, (size_t)m * k));
  GPK_TRY(sc.get(&dB, (size_t)n * k));
  GPK_TRY(sc.get(&dC, (size_t)m * n));
  GPK_HIP(hipMemcpyAsync(dA, A, (size_t)m * k * 8, hipMemcpyHostToDevice, s));
  GPK_HIP(hipMemcpyAsync(dB, B, (size_t)n * k * 8, hipMemcpyHostToDevice, s));
  GPK_HIP(hipMemcpyAsync(dC, C, (size_t)m * n * 8, hipMemcpyHostToDevice, s));
  GemmNT g{};
  g.A = dA; g.lda = k; g.B = dB; g.ldb = k; g.C = dC; g.ldc = n;
  g.m = m; g.n = n; g.k = k; g.sub = sub; g.lower = lower;
  g.xcd_sq = (xcd_order && tile == 128) ? 1 : 0;  // force the XCD-aware 8x8-square tile order of the largest updates
  if (k1 > 0 && k1 < k) {
    // same product with the K range split over two separately stored operands (the paired trailing update);
    // second segment stored separately but with the same leading dimension k (columns [0, k2) used)
    const int k2 = k - k1;
    GPK_TRY(sc.get(&dA2, (size_t)m * k));
    GPK_TRY(sc.get(&dB2, (size_t)n * k));
    GPK_HIP(hipMemcpy2DAsync(dA2, (size_t)k * 8, dA + k1, (size_t)k * 8, (size_t)k2 * 8, m, hipMemcpyDeviceToDevice, s));
    GPK_HIP(hipMemcpy2DAsync(dB2, (size_t)k * 8, dB + k1, (size_t)k * 8, (size_t)k2 * 8, n, hipMemcpyDeviceToDevice, s));
    g.A2 = dA2; g.lda2 = k; g.B2 = dB2; g.ldb2 = k; g.k1 = k1;
  }
  if (tile == 16) GPK_TRY(launch_gemm_tile128(g, s));  // the chain's one-tile kernel (16x16 output tile per wave): m = n = k = 128 only
  else GPK_TRY(launch_gemm_nt(g, tile == 128, s));
  GPK_HIP(hipMemcpyAsync(C, dC, (size_t)m * n * 8, hipMemcpyDeviceToHost, s));
  GPK_HIP(hipStreamSynchronize(s));
  return 0;
}

// C (m x n) -= A (m x K) B (n x K)^T with the K range cut into nseg segments of ks[i] columns, segment i stored on its own with
// leading dimension lds[i] >= ks[i] (K = sum ks): ONE launch of the 8-wave kernel over a K-segment table.
int gpk_test_gemm_segs(const double* A, const double* B, double* C, int m, int n, const int* ks, const int* lds, int nseg, int lower) {
  GPK_TRY(rt_require());
  if (!A || !B || !C || !ks || !lds || m < 1 || n < 1 || nseg < 1 || nseg > GPK_MAXSEG) return gpk_set_err(-1, "gpk_test_gemm_segs: bad arguments");
  int K = 0;
  for (int i = 0; i < nseg; ++i) {
    if (ks[i] < 1 || lds[i] < ks[i]) return gpk_set_err(-1, "gpk_test_gemm_segs: bad segment %d", i);
    K += ks[i];
  }
  hipStream_t s = G.s_main;
  Scratch sc;
  double *dA = nullptr, *dB = nullptr, *dC = nullptr;
  GPK_TRY(sc.get(&dA, (size_t)m * K));
  GPK_TRY(sc.get(&dB, (size_t)n * K));
  GPK_TRY(sc.get(&dC, (size_t)m * n));
  GPK_HIP(hipMemcpyAsync(dA, A, (size_t)m * K * 8, hipMemcpyHostToDevice, s));
  GPK_HIP(hipMemcpyAsync(dB, B, (size_t)n * K * 8, hipMemcpyHostToDevice, s));
  GPK_HIP(hipMemcpyAsync(dC, C, (size_t)m * n * 8, hipMemcpyHostToDevice, s));
  GemmSegs sg{};
  sg.n = nseg;
  int k0 = 0;
  for (int i = 0; i < nseg; ++i) {
    double *sa = nullptr, *sb = nullptr;
    GPK_TRY(sc.get(&sa, (size_t)m * lds[i]));
    GPK_TRY(sc.get(&sb, (size_t)n * lds[i]));
    GPK_HIP(hipMemcpy2DAsync(sa, (size_t)lds[i] * 8, dA + k0, (size_t)K * 8, (size_t)ks[i] * 8, m, hipMemcpyDeviceToDevice, s));
    GPK_HIP(hipMemcpy2DAsync(sb, (size_t)lds[i] * 8, dB + k0, (size_t)K * 8, (size_t)ks[i] * 8, n, hipMemcpyDeviceToDevice, s));
    sg.s[i] = GemmSeg{sa, sb, lds[i], lds[i], ks[i], 0};
    k0 += ks[i];
  }
  GemmNT g{};
  g.C = dC; g.ldc = n; g.m = m; g.n = n; g.sub = 1; g.lower = lower;
  GPK_TRY(launch_gemm_nt_segs(g, sg, s));
  GPK_HIP(hipMemcpyAsync(C, dC, (size_t)m * n * 8, hipMemcpyDeviceToHost, s));
  GPK_HIP(hipStreamSynchronize(s));
  return 0;
}

// One chain step of the tile pipeline on a 256 x 256 block B (lower tiles (0,0), (1,0), (1,1) used): leaf of tile (0,0), L(1,0) = B(1,0)
// Dinv^T, B(1,1) -= L(1,0) L(1,0)^T.  mode 0: the three launches (full-tile leaf, two one-tile GEMMs); 1: potf2_chain_kernel without a
// flag; 2: with a flag that the update stream sets while the kernel is running; 3: with a flag nobody sets (the bounded wait must
// give up: returns -3).  L_out 256 x 128 (rows of L(0,0), then of L(1,0)), Dinv_out 128 x 128, A11_out 128 x 128.  Returns info.
int gpk_test_chain_step(const double* B, double* L_out, double* Dinv_out, double* A11_out, int mode) {
  GPK_TRY(rt_require());
  if (!B || !L_out || !Dinv_out || !A11_out || mode < 0 || mode > 3) return gpk_set_err(-1, "gpk_test_chain_step: bad arguments");
  hipStream_t sm = G.s_main, sc = G.s_panel;
  Scratch scr;
  double *dB = nullptr, *dL = nullptr, *dD = nullptr, *junk = nullptr;
  int* dinfo = nullptr;
  unsigned* dflag = nullptr;
  GPK_TRY(scr.get(&dB, (size_t)256 * 256));
  GPK_TRY(scr.get(&dL, (size_t)256 * 128));
  GPK_TRY(scr.get(&dD, (size_t)128 * 128));
  GPK_TRY(scr.get(&dinfo, (size_t)4));
  GPK_TRY(scr.get(&dflag, (size_t)4));
  GPK_TRY(scr.get(&junk, (size_t)1 << 22));
  GPK_HIP(hipMemcpyAsync(dB, B, (size_t)256 * 256 * 8, hipMemcpyHostToDevice, sm));
  GPK_HIP(hipMemsetAsync(dinfo, 0, 16, sm));
  GPK_HIP(hipMemsetAsync(dflag, 0, 16, sm));
  GPK_HIP(hipStreamSynchronize(sm));
  if (mode == 0) {
    GPK_TRY(launch_potf2_trtri_128(dB, 256, dL, 128, dD, dinfo, 0, 0, sc, 1));
    GemmNT t1{};
    t1.A = dB + 128 * 256; t1.lda = 256; t1.B = dD; t1.ldb = 128; t1.C = dL + 128 * 128; t1.ldc = 128; t1.m = t1.n = t1.k = 128;
    GPK_TRY(launch_gemm_tile128(t1, sc));
    GemmNT t2{};
    t2.A = dL + 128 * 128; t2.lda = 128; t2.B = t2.A; t2.ldb = 128; t2.C = dB + 128 * 256 + 128; t2.ldc = 256; t2.m = t2.n = t2.k = 128; t2.sub = 1; t2.lower = 1;
    GPK_TRY(launch_gemm_tile128(t2, sc));
  } else {
    GPK_TRY(launch_potf2_chain(dB, 256, dL, 128, dD, dinfo, 0, mode == 1 ? nullptr : dflag, 7u, sc));
    if (mode == 2) {  // the flag arrives from the other stream, behind some work, while the kernel runs
      GPK_TRY(launch_fill(junk, (size_t)1 << 22, 1.0, sm));
      GPK_TRY(launch_flag_set(dflag, 7u, sm));
    }
  }
  GPK_HIP(hipStreamSynchronize(sc));
  GPK_HIP(hipStreamSynchronize(sm));
  int hinfo = 0;
  GPK_HIP(hipMemcpy(&hinfo, dinfo, 4, hipMemcpyDeviceToHost));
  GPK_HIP(hipMemcpy(L_out, dL, (size_t)256 * 128 * 8, hipMemcpyDeviceToHost));
  GPK_HIP(hipMemcpy(Dinv_out, dD, (size_t)128 * 128 * 8, hipMemcpyDeviceToHost));
  GPK_HIP(hipMemcpy2DAsync(A11_out, 128 * 8, dB + 128 * 256 + 128, 256 * 8, 128 * 8, 128, hipMemcpyDeviceToHost, sm));
  GPK_HIP(hipStreamSynchronize(sm));
  GPK_TRY(info_timeout_rc(hinfo));
  return hinfo;
}

int gpk_test_potf2_128(const double* A, double* L_out, double* Dinv_out, int full_tile) {
  GPK_TRY(rt_require());
  hipStream_t s = G.s_main;
  double *dA = nullptr, *dL = nullptr, *dD = nullptr;
  int* dinfo = nullptr;
  GPK_TRY(dev_alloc_t(&dA, 128 * 128));
  GPK_TRY(dev_alloc_t(&dL, 128 * 128));
  GPK_TRY(dev_alloc_t(&dD, 128 * 128));
  GPK_TRY(dev_alloc_t(&dinfo, 4));
  int rc = 0, hinfo = 0;
  do {
    if (hipMemcpyAsync(dA, A, 128 * 128 * 8, hipMemcpyHostToDevice, s) != hipSuccess) { rc = gpk_set_err(-2, "H2D failed"); break; }
    if (hipMemsetAsync(dinfo, 0, 16, s) != hipSuccess) { rc = gpk_set_err(-2, "memset failed"); break; }
    rc = launch_potf2_trtri_128(dA, 128, dL, 128, dD, dinfo, 0, 0, s, full_tile ? 1 : 0);
    if (rc) break;
    if (hipMemcpyAsync(L_out, dL, 128 * 128 * 8, hipMemcpyDeviceToHost, s) != hipSuccess) { rc = gpk_set_err(-2, "D2H failed"); break; }
    if (hipMemcpyAsync(Dinv_out, dD, 128 * 128 * 8, hipMemcpyDeviceToHost, s) != hipSuccess) { rc = gpk_set_err(-2, "D2H failed"); break; }
    if (hipMemcpyAsync(&hinfo, dinfo, 4, hipMemcpyDeviceToHost, s) != hipSuccess) { rc = gpk_set_err(-2, "D2H failed"); break; }
    if (hipStreamSynchronize(s) != hipSuccess) { rc = gpk_set_err(-2, "test_potf2: %s", hipGetErrorString(hipGetLastError())); break; }
    rc = hinfo;
  } while (0);
  dev_free(dA, 128 * 128 * 8);
  dev_free(dL, 128 * 128 * 8);
  dev_free(dD, 128 * 128 * 8);
  dev_free(dinfo, 16);
  return rc;
}

}  // extern "C"

extern "C" int gpk_test_dist_fit(const double* X, int N, int d, const double* log_hyp, double jitter, int P, double* L_out) {
  // as gpk_test_dist_potrf, but every virtual rank builds ITS block columns of K from X (build_K), which
  // exercises the global-column bookkeeping of the kernel-matrix builder.
  GPK_TRY(rt_require());
  if (!X || !log_hyp || !L_out || N < 1 || P < 1 || P > 16) return gpk_set_err(-1, "gpk_test_dist_fit: bad arguments");
  hipStream_t s = G.s_main;
  VirtualJob job;
  GPK_TRY(job.create(N, d, P));
  for (gpk_model* m : job.ms) {
    GPK_TRY(decode_hyp(log_hyp, d, jitter, m));
    GPK_HIP(hipMemcpyAsync(m->dX, X, (size_t)N * d * 8, hipMemcpyHostToDevice, s));
  }
  GPK_TRY(P == 1 ? fit_device(job.ms[0]) : fit_group(job.ms.data(), P));
  return model_unpack_L(job.ms[P - 1], L_out);
}

// The whole posterior of a P-rank job (fit + sharded prediction + emulated all-gather) on P virtual ranks.
// nlml_out[P], mean_out / var_out [P][M]: what every rank ends up with (they must all agree).
extern "C" int gpk_test_dist_posterior(const double* X, const double* y, int N, int d, const double* log_hyp,
                                       const double* Z, int M, int P, double* nlml_out, double* mean_out,
                                       double* var_out) {
  GPK_TRY(rt_require());
  if (!X || !y || !log_hyp || !Z || !nlml_out || !mean_out || !var_out || N < 1 || M < 1 || P < 2 || P > 16)
    return gpk_set_err(-1, "gpk_test_dist_posterior: bad arguments");
  hipStream_t s = G.s_main;
  VirtualJob job;
  GPK_TRY(job.create(N, d, P));
  double* dZ = nullptr;
  GPK_TRY(dev_alloc_t(&dZ, (size_t)M * d));
  auto body = [&]() -> int {
    GPK_HIP(hipMemcpyAsync(dZ, Z, (size_t)M * d * 8, hipMemcpyHostToDevice, s));
    for (gpk_model* m : job.ms) {
      GPK_TRY(decode_hyp(log_hyp, d, 0.0, m));
      m->has_y = true;
      GPK_HIP(hipMemcpyAsync(m->dX, X, (size_t)N * d * 8, hipMemcpyHostToDevice, s));
      GPK_HIP(hipMemsetAsync(m->dy, 0, (size_t)m->Npad * 8, s));
      GPK_HIP(hipMemcpyAsync(m->dy, y, (size_t)N * 8, hipMemcpyHostToDevice, s));
    }
    std::vector<double*> hm(P), hv(P);
    for (int r = 0; r < P; ++r) {
      hm[r] = mean_out + (size_t)r * M;
      hv[r] = var_out + (size_t)r * M;
    }
    // option fuse_predict (default): fit + sharded prediction in one go, the substitutions behind the panel events -- what
    // gpk_fit_predict / gpk_bench_posterior run on a real rank; 0: fit, then the separate sharded prediction (gpk_fit + gpk_predict)
    if (G.fuse_predict) {
      GPK_TRY(fit_predict_group(job.ms.data(), P, dZ, M, hm.data(), hv.data()));
    } else {
      GPK_TRY(fit_group(job.ms.data(), P));
      GPK_TRY(predict_group(job.ms.data(), P, dZ, M, hm.data(), hv.data()));
    }
    for (int r = 0; r < P; ++r) nlml_out[r] = job.ms[r]->nlml;
    return 0;
  };
  int rc = body();
  if (rc) (void)hipStreamSynchronize(s);
  dev_free(dZ, (size_t)M * d * 8);
  return rc;
}

#ifndef GPK_HOSTSIM  // device-only probes (hardware registers, inline asm): not part of the host-side sanitizer build
// LDS banking probe: every lane issues `reps` ds_read_b64 from one fixed address given by `pattern`; run under
// `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE` (scripts/lds_bank_probe.sh).  fr = lane & 15, fg = lane >> 4 as in
// the MFMA operand layout.  0: lane * 8 (contiguous); 1: the LDS-DMA image of the tile kernel (128-byte rows, 16-byte chunks
// XOR-swizzled by (row >> 1) & 7, k = fg); 2: rows 18 doubles apart (the register-staged image); 3: rows 17 doubles apart;
// 4: 128-byte rows with an 8-byte swizzle ((fr & 7) * 2 + ((fr >> 3) ^ (fg & 1))); 5: pattern 1 without the swizzle.
template <int PAT>
__global__ __launch_bounds__(256) void lds_bank_probe_kernel(double* out, int reps) {
  __shared__ double buf[64 * 18 + 64];
  for (int e = threadIdx.x; e < 64 * 18 + 64; e += 256) buf[e] = (double)e;
  __syncthreads();
  const int lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
  int idx;
  if (PAT == 0) idx = lane;
  else if (PAT == 1) idx = fr * 16 + ((((fg >> 1) ^ ((fr >> 1) & 7)) * 2) + (fg & 1));
  else if (PAT == 2) idx = fr * 18 + fg;
  else if (PAT == 3) idx = fr * 17 + fg;
  else if (PAT == 4) idx = fr * 16 + (((fr & 7) * 2 + ((fr >> 3) ^ (fg & 1))) ^ ((fg >> 1) * 2)) % 16;
  else idx = fr * 16 + fg;
  const double* p = buf + idx;
  double acc = 0;
  for (int r = 0; r < reps; ++r) {
    double v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(unsigned long)(__attribute__((address_space(3))) const double*)p) : "memory");
    acc += v;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

extern "C" {

// How many streams run kernels truly side by side?  nstreams streams (priority of stream i: bit i of high_mask -> highest, else normal;
// bit i of low_mask -> lowest) each get `reps` launches of a small spinning kernel (wgs workgroups, ~us_each microseconds), queued round
// robin; out_us[i] = time per launch on stream i, out_us[nstreams] = wall time of everything per launch-round.
__global__ void spin_kernel(unsigned long long ticks, unsigned* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
  if (sink && threadIdx.x == 0 && blockIdx.x == 0xffffffffu) *sink = 1;
}
int gpk_bench_streams(int nstreams, int high_mask, int low_mask, int wgs, int us_each, int reps, int use_lib_streams, double* out_us) {
  GPK_TRY(rt_require());
  if (nstreams < 1 || nstreams > 8 || reps < 1) return gpk_set_err(-1, "gpk_bench_streams: bad arguments");
  int lo = 0, hi = 0;
  GPK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  std::vector<hipStream_t> st(nstreams, nullptr);
  std::vector<bool> own(nstreams, true);
  hipStream_t lib[3] = {G.s_main, G.s_panel, G.s_pred};
  for (int i = 0; i < nstreams; ++i) {
    if ((use_lib_streams & 1) && i < 3 && lib[i]) { st[i] = lib[i]; own[i] = false; continue; }
    if ((use_lib_streams & 2) && i == nstreams - 1) { st[i] = nullptr; own[i] = false; continue; }  // the LAST stream is the null stream
    const int pr = (high_mask >> i) & 1 ? hi : ((low_mask >> i) & 1 ? lo : (lo + hi) / 2);
    GPK_HIP(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, pr));
  }
  std::vector<hipEvent_t> e0(nstreams), e1(nstreams);
  for (int i = 0; i < nstreams; ++i) { GPK_HIP(hipEventCreate(&e0[i])); GPK_HIP(hipEventCreate(&e1[i])); }
  for (int i = 0; i < nstreams; ++i) hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, st[i], 100ull, nullptr);  // queues exist
  GPK_HIP(hipDeviceSynchronize());
  const auto w0 = std::chrono::steady_clock::now();
  for (int i = 0; i < nstreams; ++i) GPK_HIP(hipEventRecord(e0[i], st[i]));
  for (int r = 0; r < reps; ++r)
    for (int i = 0; i < nstreams; ++i) hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, st[i], (unsigned long long)us_each * 100ull, nullptr);
  for (int i = 0; i < nstreams; ++i) GPK_HIP(hipEventRecord(e1[i], st[i]));
  GPK_HIP(hipDeviceSynchronize());
  out_us[nstreams] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count() / reps;
  for (int i = 0; i < nstreams; ++i) {
    float ms = 0;
    GPK_HIP(hipEventElapsedTime(&ms, e0[i], e1[i]));
    out_us[i] = 1e3 * ms / reps;
    (void)hipEventDestroy(e0[i]);
    (void)hipEventDestroy(e1[i]);
    if (own[i]) (void)hipStreamDestroy(st[i]);
  }
  return 0;
}

// the product's stream-layout probe (streams_abreast, gpk_comm.hip: what gpk_comm_init runs on its four streams) on the library's three
// streams plus n_high high-priority and n_normal normal-priority ones created here; out2 = worst, best time per launch in us
int gpk_test_streams_abreast(int n_high, int n_normal, int reps, double* out2) {
  GPK_TRY(rt_require());
  if (n_high < 0 || n_normal < 0 || n_high + n_normal > 5 || reps < 1) return gpk_set_err(-1, "gpk_test_streams_abreast: bad arguments");
  int lo = 0, hi = 0;
  GPK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  std::vector<hipStream_t> st = {G.s_main, G.s_panel, G.s_pred};
  for (int i = 0; i < n_high + n_normal; ++i) {
    hipStream_t x = nullptr;
    GPK_HIP(hipStreamCreateWithPriority(&x, hipStreamNonBlocking, i < n_high ? hi : (lo + hi) / 2));
    st.push_back(x);
  }
  const int rc = streams_abreast(st.data(), (int)st.size(), reps, out2, out2 + 1);
  for (size_t i = 3; i < st.size(); ++i) (void)hipStreamDestroy(st[i]);
  return rc;
}

int gpk_bench_lds_bank_probe(int pattern, int reps) {
  GPK_TRY(rt_require());
  Scratch sc;
  double* d = nullptr;
  GPK_TRY(sc.get(&d, (size_t)256 * 256));
  hipStream_t s = G.s_main;
  switch (pattern) {
    case 0: hipLaunchKernelGGL(lds_bank_probe_kernel<0>, dim3(256), dim3(256), 0, s, d, reps); break;
    case 1: hipLaunchKernelGGL(lds_bank_probe_kernel<1>, dim3(256), dim3(256), 0, s, d, reps); break;
    case 2: hipLaunchKernelGGL(lds_bank_probe_kernel<2>, dim3(256), dim3(256), 0, s, d, reps); break;
    case 3: hipLaunchKernelGGL(lds_bank_probe_kernel<3>, dim3(256), dim3(256), 0, s, d, reps); break;
    case 4: hipLaunchKernelGGL(lds_bank_probe_kernel<4>, dim3(256), dim3(256), 0, s, d, reps); break;
    default: hipLaunchKernelGGL(lds_bank_probe_kernel<5>, dim3(256), dim3(256), 0, s, d, reps); break;
  }
  GPK_HIP(hipGetLastError());
  GPK_HIP(hipStreamSynchronize(s));
  return 0;
}

int gpk_bench_leaf(int dbg_skip, int reps, double* us_out) {
  GPK_TRY(rt_require());
  hipStream_t s = G.s_main;
  double *dA = nullptr, *dL = nullptr, *dD = nullptr;
  int* dinfo = nullptr;
  GPK_TRY(dev_alloc_t(&dA, 128 * 128));
  GPK_TRY(dev_alloc_t(&dL, 128 * 128));
  GPK_TRY(dev_alloc_t(&dD, 128 * 128));
  GPK_TRY(dev_alloc_t(&dinfo, 4));
  std::vector<double> h(128 * 128);
  for (int i = 0; i < 128; ++i)
    for (int j = 0; j < 128; ++j) h[i * 128 + j] = (i == j ? 4.0 : 0.0) + 1.0 / (1.0 + (i > j ? i - j : j - i));
  GPK_HIP(hipMemcpyAsync(dA, h.data(), 128 * 128 * 8, hipMemcpyHostToDevice, s));
  GPK_HIP(hipMemsetAsync(dinfo, 0, 16, s));
  const int fat = (dbg_skip >> 8) & 1;  // bit 8: the full-tile form
  dbg_skip &= 0xff;
  int rc = launch_potf2_trtri_128(dA, 128, dL, 128, dD, dinfo, 0, dbg_skip, s, fat);
  hipEvent_t e0, e1;
  GPK_HIP(hipEventCreate(&e0));
  GPK_HIP(hipEventCreate(&e1));
  GPK_HIP(hipEventRecord(e0, s));
  for (int r = 0; r < reps && !rc; ++r) rc = launch_potf2_trtri_128(dA, 128, dL, 128, dD, dinfo, 0, dbg_skip, s, fat);
  GPK_HIP(hipEventRecord(e1, s));
  GPK_HIP(hipEventSynchronize(e1));
  float ms = 0;
  GPK_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (us_out) *us_out = 1e3 * ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  dev_free(dA, 128 * 128 * 8);
  dev_free(dL, 128 * 128 * 8);
  dev_free(dD, 128 * 128 * 8);
  dev_free(dinfo, 16);
  return rc;
}

// leaf kernel on the chain stream while the update stream keeps the chip busy (reps bit 24: the full-tile form)
int gpk_bench_leaf_contended(int reps, double* us_alone, double* us_contended) {
  const int fat = (reps >> 24) & 1;
  const int dbg = (reps >> 16) & 0xff;  // leaf dbg_skip mode
  reps &= 0xffff;
  GPK_TRY(rt_require());
  hipStream_t sl = G.s_panel ? G.s_panel : G.s_main, sm = G.s_main;
  double *dA = nullptr, *dL = nullptr, *dD = nullptr, *gA = nullptr, *gC = nullptr;
  int* dinfo = nullptr;
  const int gm = 8192, gk = 256;
  GPK_TRY(dev_alloc_t(&dA, 128 * 128));
  GPK_TRY(dev_alloc_t(&dL, 128 * 128));
  GPK_TRY(dev_alloc_t(&dD, 128 * 128));
  GPK_TRY(dev_alloc_t(&dinfo, 4));
  GPK_TRY(dev_alloc_t(&gA, (size_t)gm * gk));
  GPK_TRY(dev_alloc_t(&gC, (size_t)gm * gm));
  std::vector<double> h(128 * 128);
  for (int i = 0; i < 128; ++i)
    for (int j = 0; j < 128; ++j) h[i * 128 + j] = (i == j ? 4.0 : 0.0) + 1.0 / (1.0 + (i > j ? i - j : j - i));
  GPK_HIP(hipMemcpy(dA, h.data(), 128 * 128 * 8, hipMemcpyHostToDevice));
  GPK_HIP(hipMemset(dinfo, 0, 16));
  GPK_TRY(launch_fill(gA, (size_t)gm * gk, 1e-3, sm));
  GPK_TRY(launch_fill(gC, (size_t)gm * gm, 1.0, sm));
  GPK_HIP(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  GPK_HIP(hipEventCreate(&e0));
  GPK_HIP(hipEventCreate(&e1));
  auto run_leaves = [&](double* out) -> int {
    GPK_HIP(hipEventRecord(e0, sl));
    for (int r = 0; r < reps; ++r) GPK_TRY(launch_potf2_trtri_128(dA, 128, dL, 128, dD, dinfo, 0, dbg, sl, fat));
    GPK_HIP(hipEventRecord(e1, sl));
    GPK_HIP(hipEventSynchronize(e1));
    float ms = 0;
    GPK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *out = 1e3 * ms / reps;
    int hi[8] = {0};
    GPK_HIP(hipMemcpy(hi, dinfo, 32, hipMemcpyDeviceToHost));
    if (dbg & 8) printf("   phases (us): load %.1f | eliminate+scale+L store %.1f | inverse %.1f | Dinv store issue %.1f\n", (hi[4]-hi[3])*0.01, (hi[5]-hi[4])*0.01, (hi[6]-hi[5])*0.01, (hi[7]-hi[6])*0.01);
    if (dbg & 8) printf("   elimination: %d shader cycles, %d ticks of 10 ns -> %.3f GHz, %.1f us\n", hi[1], hi[2], hi[2] ? hi[1] / (hi[2] * 10.0) : 0.0, hi[2] * 0.01);
    return 0;
  };
  GPK_TRY(run_leaves(us_alone));
  GemmNT g{};
  g.A = gA; g.lda = gk; g.B = gA; g.ldb = gk; g.C = gC; g.ldc = gm; g.m = gm; g.n = gm; g.k = gk; g.sub = 1; g.lower = 1;
  for (int r = 0; r < 120; ++r) GPK_TRY(launch_gemm_nt(g, true, sm));  // ~15 ms of back-to-back trailing updates
  GPK_TRY(run_leaves(us_contended));
  GPK_HIP(hipDeviceSynchronize());
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  dev_free(dA, 128 * 128 * 8); dev_free(dL, 128 * 128 * 8); dev_free(dD, 128 * 128 * 8); dev_free(dinfo, 16);
  dev_free(gA, (size_t)gm * gk * 8); dev_free(gC, (size_t)gm * gm * 8);
  return 0;
}

// Feasibility probe for a tile-level critical chain: `reps` steps of [leaf, `nsmall` one-tile-row GEMMs (m = 128, n = 128,
// k = 128, 64x64 tiles = 4 workgroups each)] on the panel-chain stream, alone and while the update stream runs
// back-to-back trailing updates.  us per step.
int gpk_bench_chain_step(int reps, int nsmall, int small_m, double* us_alone, double* us_contended) {
  GPK_TRY(rt_require());
  hipStream_t sl = G.s_panel, sm = G.s_main;
  double *dA = nullptr, *dL = nullptr, *dD = nullptr, *gA = nullptr, *gC = nullptr, *tC = nullptr;
  int* dinfo = nullptr;
  const int gm = 8192, gk = 256;
  GPK_TRY(dev_alloc_t(&dA, 128 * 128));
  GPK_TRY(dev_alloc_t(&dL, 128 * 128));
  GPK_TRY(dev_alloc_t(&dD, 128 * 128));
  GPK_TRY(dev_alloc_t(&dinfo, 4));
  GPK_TRY(dev_alloc_t(&gA, (size_t)gm * gk));
  GPK_TRY(dev_alloc_t(&gC, (size_t)gm * gm));
  GPK_TRY(dev_alloc_t(&tC, (size_t)small_m * 128));
  std::vector<double> h(128 * 128);
  for (int i = 0; i < 128; ++i)
    for (int j = 0; j < 128; ++j) h[i * 128 + j] = (i == j ? 4.0 : 0.0) + 1.0 / (1.0 + (i > j ? i - j : j - i));
  GPK_HIP(hipMemcpy(dA, h.data(), 128 * 128 * 8, hipMemcpyHostToDevice));
  GPK_HIP(hipMemset(dinfo, 0, 16));
  GPK_TRY(launch_fill(gA, (size_t)gm * gk, 1e-3, sm));
  GPK_TRY(launch_fill(gC, (size_t)gm * gm, 1.0, sm));
  GPK_TRY(launch_fill(tC, (size_t)small_m * 128, 1.0, sm));
  GPK_HIP(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  GPK_HIP(hipEventCreate(&e0));
  GPK_HIP(hipEventCreate(&e1));
  auto run_chain = [&](double* out) -> int {
    GPK_HIP(hipEventRecord(e0, sl));
    for (int r = 0; r < reps; ++r) {
      GPK_TRY(leaf_launch(dA, 128, dL, 128, dD, dinfo, 0, sl));
      for (int q = 0; q < nsmall; ++q) {
        GemmNT g{};
        g.A = gA; g.lda = gk; g.B = dD; g.ldb = 128; g.C = tC; g.ldc = 128;
        g.m = small_m; g.n = 128; g.k = 128; g.sub = (q & 1);
        GPK_TRY(launch_gemm_nt(g, false, sl));
      }
    }
    GPK_HIP(hipEventRecord(e1, sl));
    GPK_HIP(hipEventSynchronize(e1));
    float ms = 0;
    GPK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *out = 1e3 * ms / reps;
    return 0;
  };
  {
    GPK_TRY(run_chain(us_alone));
    GemmNT g{};
    g.A = gA; g.lda = gk; g.B = gA; g.ldb = gk; g.C = gC; g.ldc = gm; g.m = gm; g.n = gm; g.k = gk; g.sub = 1; g.lower = 1;
    const int nbig = (int)(reps * 0.12) + 20;  // ~0.37 ms each: enough to outlast the chain
    for (int r = 0; r < nbig; ++r) GPK_TRY(launch_gemm_nt(g, true, sm));
    GPK_TRY(run_chain(us_contended));
  }
  GPK_HIP(hipDeviceSynchronize());
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  dev_free(dA, 128 * 128 * 8); dev_free(dL, 128 * 128 * 8); dev_free(dD, 128 * 128 * 8); dev_free(dinfo, 16);
  dev_free(gA, (size_t)gm * gk * 8); dev_free(gC, (size_t)gm * gm * 8); dev_free(tC, (size_t)small_m * 128 * 8);
  return 0;
}

// device-resident timing of the kernel-matrix builder: mode 0 SE, 1 squared distance, 2 stores only
int gpk_bench_se_build(int N, int d, int mode, int lower, int reps, double* ms_out) {
  GPK_TRY(rt_require());
  hipStream_t s = G.s_main;
  const int np_ = round_up(N, 128);
  double *dX = nullptr, *dK = nullptr;
  GPK_TRY(dev_alloc_t(&dX, (size_t)N * d));
  GPK_TRY(dev_alloc_t(&dK, (size_t)np_ * np_));
  std::vector<double> h((size_t)N * d);
  unsigned long long st = 88172645463325252ull;
  for (auto& v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (double)(st >> 11) / 9007199254740992.0; }
  GPK_HIP(hipMemcpyAsync(dX, h.data(), h.size() * 8, hipMemcpyHostToDevice, s));
  SEParams se{};
  se.d = d; se.sf2 = 1.0; se.diag_add = 0.01;
  for (int k = 0; k < d; ++k) se.ell[k] = 0.7;
  double* dS = nullptr;  // mode 3: the SE build in its scalar-operand form
  const bool skip = mode == 4;  // ... without the pre-pass (timing only)
  if (mode == 3 || mode == 4) {
    GPK_TRY(dev_alloc_t(&dS, se_scratch_doubles(np_, np_, d, false)));
    mode = 0;
  }
  SEScalarForm form;
  form.scratch = dS;
  int rc = launch_se_build(dX, N, np_, dX, N, np_, se, 1, mode, dK, np_, lower, s, 0, nullptr, form);
  form.rows_ready = skip ? 1 : 0;
  hipEvent_t e0, e1;
  GPK_HIP(hipEventCreate(&e0));
  GPK_HIP(hipEventCreate(&e1));
  GPK_HIP(hipEventRecord(e0, s));
  for (int r = 0; r < reps && !rc; ++r) rc = launch_se_build(dX, N, np_, dX, N, np_, se, 1, mode, dK, np_, lower, s, 0, nullptr, form);
  GPK_HIP(hipEventRecord(e1, s));
  GPK_HIP(hipEventSynchronize(e1));
  float ms = 0;
  GPK_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (ms_out) *ms_out = ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  dev_free(dX, (size_t)N * d * 8);
  dev_free(dK, (size_t)np_ * np_ * 8);
  if (dS) dev_free(dS, se_scratch_doubles(np_, np_, d, false) * 8);
  return rc;
}

// One rank of a P-rank block-cyclic posterior ALONE on the GPU: the whole virtual job runs once (so that every rank holds
// every panel of the true factor), then rank r's share is replayed `reps` times by itself -- its K blocks, its updates, the
// panels it owns (incl. the tile pipeline if it owns the tail block), its shard of the prediction behind the panel events;
// panels of other owners arrive as device copies from their models (the broadcasts' stand-in), never late.  ms_out[r]: the
// time rank r needs when it never waits for a peer -- a LOWER bound of the job's time on P GPUs (max over r), and what one
// GPU can measure of it.  ms_out[P]: the one-GPU posterior of the same problem in the same process.
// comm_stream != 0: the panels of other owners arrive on a high-priority COMMUNICATION stream of their own, as on a real rank (update,
// chain, substitution and communication stream: four), instead of sharing the substitution stream as in the test harness.
int gpk_bench_dist_rank_alone(const double* X, const double* y, int N, int d, const double* log_hyp, const double* Z, int M, int P,
                              int reps, int comm_stream, double* ms_out) {
  GPK_TRY(rt_require());
  if (!X || !y || !log_hyp || !Z || !ms_out || N < 1 || M < 1 || P < 2 || P > 16 || reps < 1)
    return gpk_set_err(-1, "gpk_bench_dist_rank_alone: bad arguments");
  if (comm_stream && G.s_comm) return gpk_set_err(-1, "gpk_bench_dist_rank_alone: the process has a communication stream already");
  struct TempComm {  // a real rank's fourth stream for the duration of the call
    bool on = false;
    ~TempComm() {
      if (!on) return;
      (void)hipStreamSynchronize(G.s_comm);
      (void)hipStreamDestroy(G.s_comm);
      G.s_comm = nullptr;
    }
  } tc;
  if (comm_stream) {
    int lo = 0, hi = 0;
    GPK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    GPK_HIP(hipStreamCreateWithPriority(&G.s_comm, hipStreamNonBlocking, hi));
    tc.on = true;
  }
  hipStream_t s = G.s_main;
  VirtualJob job;
  GPK_TRY(job.create(N, d, P));
  double* dZ = nullptr;
  GPK_TRY(dev_alloc_t(&dZ, (size_t)M * d));
  std::vector<double> hm((size_t)P * M), hv((size_t)P * M);
  auto body = [&]() -> int {
    GPK_HIP(hipMemcpyAsync(dZ, Z, (size_t)M * d * 8, hipMemcpyHostToDevice, s));
    for (gpk_model* m : job.ms) {
      GPK_TRY(decode_hyp(log_hyp, d, 0.0, m));
      m->has_y = true;
      GPK_HIP(hipMemcpyAsync(m->dX, X, (size_t)N * d * 8, hipMemcpyHostToDevice, s));
      GPK_HIP(hipMemsetAsync(m->dy, 0, (size_t)m->Npad * 8, s));
      GPK_HIP(hipMemcpyAsync(m->dy, y, (size_t)N * 8, hipMemcpyHostToDevice, s));
    }
    std::vector<double*> pm(P), pv(P);
    for (int r = 0; r < P; ++r) { pm[r] = hm.data() + (size_t)r * M; pv[r] = hv.data() + (size_t)r * M; }
    GPK_TRY(fit_predict_group(job.ms.data(), P, dZ, M, pm.data(), pv.data()));
    const double nl0 = job.ms[0]->nlml;
    for (int r = 0; r < P; ++r) {
      gpk_model* one = job.ms[r];
      double best = 1e300;
      for (int k = 0; k < reps + 1; ++k) {  // (first replay: warm-up)
        GPK_HIP(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        GPK_TRY(fit_predict_group(&one, 1, dZ, M, &pm[r], &pv[r]));  // (ends with the read-back of the gathered prediction: synchronous)
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (k > 0 && ms < best) best = ms;
        if (one->nlml != nl0) return gpk_set_err(-1, "internal: rank %d replayed alone disagrees with the group (%.17g vs %.17g)", r, one->nlml, nl0);
      }
      ms_out[r] = best;
    }
    return 0;
  };
  int rc = body();
  if (rc) (void)hipStreamSynchronize(s);
  dev_free(dZ, (size_t)M * d * 8);
  return rc;
}

int gpk_bench_gemm_nt(int m, int n, int k, int lower, int tile, int reps, double* ms_out) {
  GPK_TRY(rt_require());
  hipStream_t s = G.s_main;
  double *dA = nullptr, *dB = nullptr, *dC = nullptr;
  GPK_TRY(dev_alloc_t(&dA, (size_t)m * k));
  GPK_TRY(dev_alloc_t(&dC, (size_t)m * n));
  if (!lower) GPK_TRY(dev_alloc_t(&dB, (size_t)n * k));
  // random-ish operands (DVFS: zero operands read high, guide rule 25)
  std::vector<double> h((size_t)std::max(m, n) * k);
  unsigned long long st = 88172645463325252ull;
  for (auto& v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = ((double)(st >> 11) / 9007199254740992.0 - 0.5) * 1e-3; }
  GPK_HIP(hipMemcpyAsync(dA, h.data(), (size_t)m * k * 8, hipMemcpyHostToDevice, s));
  if (!lower) GPK_HIP(hipMemcpyAsync(dB, h.data(), (size_t)n * k * 8, hipMemcpyHostToDevice, s));
  GPK_TRY(launch_fill(dC, (size_t)m * n, 1.0, s));
  GemmNT g{};
  g.A = dA; g.lda = k; g.B = lower ? dA : dB; g.ldb = k; g.C = dC; g.ldc = n;
  g.m = m; g.n = n; g.k = k; g.sub = 1; g.lower = lower & 1;
  auto go = [&]() { return tile == 64 ? launch_gemm_nt(g, false, s) : launch_gemm_auto(g, s); };
#ifdef GPK_GEMM_DBG_VARIANTS
  // GPK_GEMM_TRACE=<file> with GPK_GEMM_DBG=32: per-workgroup stamps of ONE launch (after a warm-up launch) as raw u64[8] records
  extern unsigned long long* g_gemm_trace;
  if (getenv("GPK_GEMM_TRACE") && tile != 64) {
    const size_t nw = (size_t)8 * 65536;
    unsigned long long* dtr = nullptr;
    GPK_HIP(hipMalloc((void**)&dtr, nw * 8));
    GPK_HIP(hipMemset(dtr, 0, nw * 8));
    (void)go();
    GPK_HIP(hipStreamSynchronize(s));
    g_gemm_trace = dtr;
    (void)go();
    GPK_HIP(hipStreamSynchronize(s));
    g_gemm_trace = nullptr;
    std::vector<unsigned long long> h(nw);
    GPK_HIP(hipMemcpy(h.data(), dtr, nw * 8, hipMemcpyDeviceToHost));
    (void)hipFree(dtr);
    if (FILE* f = fopen(getenv("GPK_GEMM_TRACE"), "wb")) {
      fwrite(h.data(), 8, nw, f);
      fclose(f);
    }
  }
#endif
#ifdef GPK_GEMM_DBG_VARIANTS
  // GPK_GEMM_TRACE=<file> with GPK_GEMM_DBG=32: per-workgroup stamps of ONE launch (after a warm-up launch) as raw u64[5] records
  extern unsigned long long* g_gemm_trace;
  if (getenv("GPK_GEMM_TRACE") && tile != 64) {
    const size_t nw = (size_t)5 * 65536;
    unsigned long long* dtr = nullptr;
    GPK_HIP(hipMalloc((void**)&dtr, nw * 8));
    GPK_HIP(hipMemset(dtr, 0, nw * 8));
    (void)go();
    GPK_HIP(hipStreamSynchronize(s));
    g_gemm_trace = dtr;
    (void)go();
    GPK_HIP(hipStreamSynchronize(s));
    g_gemm_trace = nullptr;
    std::vector<unsigned long long> h(nw);
    GPK_HIP(hipMemcpy(h.data(), dtr, nw * 8, hipMemcpyDeviceToHost));
    (void)hipFree(dtr);
    if (FILE* f = fopen(getenv("GPK_GEMM_TRACE"), "wb")) {
      fwrite(h.data(), 8, nw, f);
      fclose(f);
    }
  }
#endif
  int rc = go();
  hipEvent_t e0, e1;
  GPK_HIP(hipEventCreate(&e0));
  GPK_HIP(hipEventCreate(&e1));
  GPK_HIP(hipEventRecord(e0, s));
  for (int r = 0; r < reps && !rc; ++r) rc = go();
  GPK_HIP(hipEventRecord(e1, s));
  GPK_HIP(hipEventSynchronize(e1));
  float ms = 0;
  GPK_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (ms_out) *ms_out = ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  dev_free(dA, (size_t)m * k * 8);
  dev_free(dC, (size_t)m * n * 8);
  if (dB) dev_free(dB, (size_t)n * k * 8);
  return rc;
}

}  // extern "C"
#endif  // GPK_HOSTSIM

#define GPK_NCCL(x)                                                                                          \
  do {                                                                                                       \
    ncclResult_t r_ = (x);                                                                                   \
    if (r_ != ncclSuccess)                                                                                   \
      return gpk_set_err(-3, "%s failed: %s (%s:%d)", #x, ncclGetErrorString(r_), __FILE__, __LINE__);      \
  } while (0)

extern "C" {

// Self-check of the RCCL calls this library makes, on a ONE-rank communicator (all a 1-GPU box can host): the same
// datatypes, counts and stream pairing as the multi-rank path -- panel-sized ncclBroadcast on a high-priority
// communication stream, ncclAllGather and ncclAllReduce(max) on the update stream -- with the payload checked afterwards.
// It cannot prove anything about links between GPUs; it does prove the calls are accepted on these streams.
int gpk_test_rccl_selfcheck(int n_doubles) {
  GPK_TRY(rt_require());
  if (G.nccl) return gpk_set_err(-1, "gpk_test_rccl_selfcheck: a communicator is already active");
  if (n_doubles < 1) return gpk_set_err(-1, "gpk_test_rccl_selfcheck: n_doubles=%d", n_doubles);
  ncclUniqueId id;
  GPK_NCCL(ncclGetUniqueId(&id));
  ncclComm_t c;
  GPK_NCCL(ncclCommInitRank(&c, 1, id, 0));
  double *a = nullptr, *b = nullptr;
  int rc = 0;
  const size_t n = (size_t)n_doubles;
  hipStream_t sc = nullptr;
  {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, hi) != hipSuccess) {
      (void)hipGetLastError();
      ncclCommDestroy(c);
      return gpk_set_err(-2, "gpk_test_rccl_selfcheck: no communication stream");
    }
  }
  if (dev_alloc_t(&a, n) || dev_alloc_t(&b, n)) rc = -2;
  std::vector<double> h(n), back(n);
  for (size_t i = 0; i < n; ++i) h[i] = (double)(i % 1000) - 499.5;
  do {
    if (rc) break;
    if (hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice) != hipSuccess) { rc = gpk_set_err(-2, "H2D failed"); break; }
    // the broadcast on a high-priority stream of its own while the all-gather / all-reduce use the update stream: the
    // stream pairing of the production path (potrf_group broadcasts on G.s_comm, fit_group / predict_group reduce on s_main)
    ncclResult_t r = ncclBroadcast(a, a, n, ncclDouble, 0, c, sc);
    if (r == ncclSuccess) r = ncclAllGather(a, b, n, ncclDouble, c, G.s_main);
    if (r == ncclSuccess) r = ncclAllReduce(b, b, n, ncclDouble, ncclMax, c, G.s_main);
    if (r != ncclSuccess) { rc = gpk_set_err(-3, "RCCL call failed: %s", ncclGetErrorString(r)); break; }
    if (hipDeviceSynchronize() != hipSuccess) { rc = gpk_set_err(-2, "sync after RCCL calls failed: %s", hipGetErrorString(hipGetLastError())); break; }
    if (hipMemcpy(back.data(), b, n * 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = gpk_set_err(-2, "D2H failed"); break; }
    for (size_t i = 0; i < n; ++i)
      if (back[i] != h[i]) { rc = gpk_set_err(-3, "payload mismatch at %zu", i); break; }
  } while (0);
  if (a) dev_free(a, n * 8);
  if (b) dev_free(b, n * 8);
  (void)hipDeviceSynchronize();
  ncclCommDestroy(c);
  (void)hipStreamDestroy(sc);
  return rc;
}

}  // extern "C"
