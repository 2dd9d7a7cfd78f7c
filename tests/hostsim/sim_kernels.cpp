// TEST INFRASTRUCTURE ONLY (tests/hostsim): plain-loop CPU stand-ins for every launch_* entry of gpk_internal.h, so
// that the HOST orchestration of libgpk (panel schedules, block-cyclic phases, pool allocator, event bookkeeping, the
// Laplace / append drivers) runs under AddressSanitizer + UBSan without a GPU.  Each stand-in touches exactly the
// memory the HIP kernel touches (same tiles, same padding rules), so a wrong size or offset computed by the host code
// is a sanitizer report here instead of a fault on the device.  Nothing in gptest_amd/ or bench.py can reach this file.
#include <unistd.h>
#include <cstdlib>
#include "gpk_internal.h"
#include <algorithm>
#include <cmath>
#include <cstring>

#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#endif

// The dot products are where the simulation spends its time.  The operand ranges are checked against AddressSanitizer's shadow
// memory as a whole (a poisoned byte is then touched by an instrumented access, which raises the usual report), the loop itself
// runs uninstrumented: the same extents are verified at a fifth of the cost.  Summation order: element by element, as before.
static inline void sim_check_range(const double* p, long n) {
#if defined(__SANITIZE_ADDRESS__)
  if (void* bad = __asan_region_is_poisoned(const_cast<double*>(p), (size_t)n * sizeof(double))) {
    volatile char c = *static_cast<volatile char*>(bad);
    (void)c;
  }
#else
  (void)p; (void)n;
#endif
}
__attribute__((no_sanitize("address", "undefined"))) static double sim_dot_raw(double acc, const double* a, const double* b, int k) {
  for (int kk = 0; kk < k; ++kk) acc += a[kk] * b[kk];
  return acc;
}
static inline double sim_dot(double acc, const double* a, const double* b, int k) {
  sim_check_range(a, k);
  sim_check_range(b, k);
  return sim_dot_raw(acc, a, b, k);
}

thread_local dim3 blockIdx, threadIdx, blockDim, gridDim;
int g_gemm_polite = 1;
unsigned* g_cu_guest = nullptr;  // the host code allocates and frees the guest table; no stand-in kernel looks at it
hipStream_t g_chain_stream = nullptr;  // set by gpk_init; only the device kernel reads what follows from it

int launch_gemm_nt(const GemmNT& g, bool tile128, hipStream_t) {
  const int BT = tile128 ? 128 : 64;
  if (g.m <= 0 || g.n <= 0) return 0;
  if (g.m % BT || g.n % BT || g.k % 16 || g.k <= 0) return gpk_set_err(-1, "gemm_nt: bad shape m=%d n=%d k=%d (tile %d)", g.m, g.n, g.k, BT);
  if ((g.lda & 1) || (g.ldb & 1)) return gpk_set_err(-1, "gemm_nt: odd leading dimension");
  if (g.lower && g.cyc_P == 0 && g.m < g.n) return gpk_set_err(-1, "gemm_nt: lower needs m >= n");
  if (g.wait_flag && (tile128 || (int)(*g.wait_flag - g.wait_val) < 0)) return gpk_set_err(-1, "hostsim: a polling launch was queued before its producer");
  if (g.k1 > 0 && (g.k1 % 16 || g.k1 >= g.k || !g.A2 || !g.B2 || (g.lda2 & 1) || (g.ldb2 & 1) || g.cyc_P != 0 || !tile128))
    return gpk_set_err(-1, "gemm_nt: bad second K segment");
  const int ntm = g.m / BT, ntn = g.n / BT;
  for (int ti = 0; ti < ntm; ++ti)
    for (int tj = 0; tj < ntn; ++tj) {
      long brow0;
      if (g.cyc_P > 0) {
        const int tpb = g.cyc_NB / BT;
        const int t = tj + g.cyc_c0 / BT;
        const long gcol = ((long)(t / tpb + g.cyc_lg0) * g.cyc_P + g.cyc_r) * g.cyc_NB + (long)(t % tpb) * BT;
        if (g.lower && gcol > g.row_g0 + (long)ti * BT) continue;
        brow0 = gcol - g.brow_g0;
      } else {
        if (g.lower && tj > ti) continue;
        brow0 = (long)tj * BT;
      }
      for (int i = 0; i < BT; ++i)
        for (int j = 0; j < BT; ++j) {
          const long ra = (long)ti * BT + i, rb = brow0 + j;
          const int ka = g.k1 > 0 ? g.k1 : g.k;
          double acc = sim_dot(0.0, g.A + ra * g.lda, g.B + rb * g.ldb, ka);
          if (g.k1 > 0) acc = sim_dot(acc, g.A2 + ra * g.lda2, g.B2 + rb * g.ldb2, g.k - g.k1);
          double* c = g.C + ra * g.ldc + (long)tj * BT + j;
          *c = g.sub ? *c - acc : acc;
        }
    }
  return 0;
}

// the 8-wave kernel over a K-segment table: the same tiles, every segment's rows read with its own leading dimension
int launch_gemm_nt_segs(const GemmNT& g, const GemmSegs& segs, hipStream_t, int persist_grid, unsigned* counter) {
  if (persist_grid > 0 && counter) *counter = 0;  // (the device form zeroes the tile counter on the stream)
  if (g.m <= 0 || g.n <= 0) return 0;
  if (g.m % 128 || g.n % 128 || g.cyc_P != 0) return gpk_set_err(-1, "gemm_nt_segs: bad shape m=%d n=%d", g.m, g.n);
  if (segs.n < 1 || segs.n > GPK_MAXSEG) return gpk_set_err(-1, "gemm_nt_segs: %d segments", segs.n);
  if (g.lower && g.m < g.n) return gpk_set_err(-1, "gemm_nt_segs: lower needs m >= n");
  for (int i = 0; i < segs.n; ++i) {
    const GemmSeg& q = segs.s[i];
    if (q.k <= 0 || q.k % 16 || !q.A || !q.B || (q.lda & 1) || (q.ldb & 1) || q.lda < q.k || q.ldb < q.k) return gpk_set_err(-1, "gemm_nt_segs: bad segment %d", i);
  }
  const int ntm = g.m / 128, ntn = g.n / 128;
  for (int ti = 0; ti < ntm; ++ti)
    for (int tj = 0; tj < ntn; ++tj) {
      if (g.lower && tj > ti) continue;
      for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) {
          const long ra = (long)ti * 128 + i, rb = (long)tj * 128 + j;
          double* c = g.C + ra * g.ldc + rb;
          double v = g.sub ? *c : 0.0;
          for (int q = 0; q < segs.n; ++q) {
            const GemmSeg& sg = segs.s[q];
            const double acc = sim_dot(0.0, sg.A + ra * sg.lda, sg.B + rb * sg.ldb, sg.k);
            v = g.sub ? v - acc : v + acc;
          }
          *c = v;
        }
    }
  return 0;
}

// the one-tile kernel of the chain: 16x16 tiles that touch the lower triangle only (lower), exactly the elements the kernel touches
int launch_gemm_tile128(const GemmNT& g, hipStream_t) {
  if (g.m != 128 || g.n != 128 || g.k != 128 || g.cyc_P != 0 || g.k1 != 0) return gpk_set_err(-1, "gemm_tile128: one 128 x 128 x 128 product only");
  for (int ti = 0; ti < 8; ++ti)
    for (int tj = 0; tj < 8; ++tj) {
      if (g.lower && tj > ti) continue;
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          const long ra = ti * 16 + i, rb = tj * 16 + j;
          const double acc = sim_dot(0.0, g.A + ra * g.lda, g.B + rb * g.ldb, 128);
          double* c = g.C + ra * g.ldc + rb;
          *c = g.sub ? *c - acc : acc;
        }
    }
  return 0;
}

int launch_potf2_trtri_128(const double* A, long lda, double* L, long ldl, double* Dinv, int* info, int col0, int, hipStream_t, int) {
  static thread_local double t[128][128], inv[128][128];
  for (int i = 0; i < 128; ++i)
    for (int j = 0; j < 128; ++j) t[i][j] = (j <= i) ? A[(long)i * lda + j] : 0.0;
  for (int j = 0; j < 128; ++j) {
    double p = t[j][j];
    for (int k = 0; k < j; ++k) p -= t[j][k] * t[j][k];
    if (!(p > 0.0)) {
      if (*info == 0) *info = col0 + j + 1;
      p = NAN;
    }
    const double l = std::sqrt(p);
    t[j][j] = l;
    for (int i = j + 1; i < 128; ++i) {
      double v = t[i][j];
      for (int k = 0; k < j; ++k) v -= t[i][k] * t[j][k];
      t[i][j] = v / l;
    }
  }
  for (int j = 0; j < 128; ++j) {
    for (int i = 0; i < 128; ++i) inv[i][j] = 0.0;
    inv[j][j] = 1.0 / t[j][j];
    for (int i = j + 1; i < 128; ++i) {
      double v = 0.0;
      for (int k = j; k < i; ++k) v -= t[i][k] * inv[k][j];
      inv[i][j] = v / t[i][i];
    }
  }
  for (int i = 0; i < 128; ++i)
    for (int j = 0; j < 128; ++j) {
      L[(long)i * ldl + j] = t[i][j];
      Dinv[i * 128 + j] = inv[i][j];
    }
  return 0;
}

// the fused chain step: the leaf, then the two one-tile products exactly as the separate launches do them
int launch_potf2_chain(double* A, long lda, double* L, long ldl, double* Dinv, int* info, int col0, const unsigned* flag, unsigned want, hipStream_t s,
                       unsigned* done_leaf, unsigned* done_t1, int do_t2) {
  if (done_leaf) *done_leaf = want;
  if (done_t1) *done_t1 = want;
  if (flag && (int)(*flag - want) < 0) return gpk_set_err(-1, "hostsim: the fused chain step was queued before the look-ahead update it waits for");
  GPK_TRY(launch_potf2_trtri_128(A, lda, L, ldl, Dinv, info, col0, 0, s, 1));
  GemmNT t1{};
  t1.A = A + 128 * lda; t1.lda = lda; t1.B = Dinv; t1.ldb = 128; t1.C = L + 128 * ldl; t1.ldc = ldl; t1.m = t1.n = t1.k = 128;
  GPK_TRY(launch_gemm_tile128(t1, s));
  if (!do_t2) return 0;
  GemmNT t2{};
  t2.A = L + 128 * ldl; t2.lda = ldl; t2.B = t2.A; t2.ldb = ldl; t2.C = A + 128 * lda + 128; t2.ldc = lda; t2.m = t2.n = t2.k = 128; t2.sub = 1; t2.lower = 1;
  return launch_gemm_tile128(t2, s);
}
int launch_flag_set(unsigned* flag, unsigned v, hipStream_t) { *flag = v; return 0; }
int launch_flag_probe(const unsigned* flag, unsigned want, unsigned* seen, hipStream_t) { (void)flag; (void)want; *seen = 1; return 0; }  // (the host simulation runs every stream in enqueue order)

// the scalar-operand form's contract (SEScalarForm): what the pre-pass would leave in the scratch -- scaled points, then norms
static void sim_scale(const double* P, int n, int n_pad, const SEParams& p, const double* centre, std::vector<double>* S) {
  S->assign((size_t)n_pad * (p.d + 1), 0.0);
  for (int i = 0; i < n; ++i) {
    double acc = 0.0;
    for (int k = 0; k < p.d; ++k) {
      const double v = (P[(long)i * p.d + k] - centre[k]) / p.ell[k] * 0.8493218002880191;
      (*S)[(size_t)i * p.d + k] = v;
      acc = std::fma(v, v, acc);
    }
    (*S)[(size_t)n_pad * p.d + i] = acc;
  }
}
int launch_se_scale(const double* P, int n, int n_pad, const SEParams& p, const double* centre, double* S, hipStream_t) {
  if (p.d < 1 || p.d > 8 || !S) return gpk_set_err(-1, "se_scale: d=%d outside 1..8 or no scratch", p.d);
  std::vector<double> v;
  sim_scale(P, n, n_pad, p, centre, &v);
  std::memcpy(S, v.data(), v.size() * 8);
  return 0;
}
int launch_se_build(const double* R, int nr, int nr_pad, const double* Cc, int nc, int nc_pad, const SEParams& p, int same_set,
                    int mode, double* out, long ld, long lower_only, hipStream_t, int col_g0, const double* centre, const SEScalarForm& form) {
  if (nr_pad % 128 || nc_pad % 128 || (ld & 1)) return gpk_set_err(-1, "se_build: padded shape must be tile aligned");
  if (p.d < 1 || p.d > 32) return gpk_set_err(-1, "se_build: input dimension d=%d outside 1..32", p.d);
  if (!centre) centre = Cc;
  if (mode == 0 && form.scratch && p.d <= 8) {
    // touch exactly what the device path touches (AddressSanitizer sees the extents) and hold the caller to the contract:
    // "rows_ready" / "cols_scaled" must really hold this call's scaled points (same p, same centre)
    std::vector<double> S;
    sim_scale(R, nr, nr_pad, p, centre, &S);
    if (form.rows_ready) {
      const double* norms = form.rows_norms ? form.rows_norms : form.scratch + (size_t)nr_pad * p.d;
      if (std::memcmp(form.scratch, S.data(), (size_t)nr_pad * p.d * 8) || std::memcmp(norms, S.data() + (size_t)nr_pad * p.d, (size_t)nr_pad * 8))
        return gpk_set_err(-1, "se_build (sim): rows_ready, but the scratch does not hold this call's scaled rows");
    } else {
      std::memcpy(form.scratch, S.data(), S.size() * 8);
    }
    const bool cols_in_rows = Cc >= R && Cc + (size_t)nc * p.d <= R + (size_t)nr * p.d && (Cc - R) % p.d == 0 && (Cc - R) / p.d + nc_pad <= nr_pad;
    if (!cols_in_rows) {
      std::vector<double> C;
      if (form.cols_scaled) {
        if (form.cols_pad < nc_pad) return gpk_set_err(-1, "se_build: %d scaled column points, %d needed", form.cols_pad, nc_pad);
        sim_scale(Cc, nc, form.cols_pad, p, centre, &C);
        if (std::memcmp(form.cols_scaled, C.data(), C.size() * 8)) return gpk_set_err(-1, "se_build (sim): cols_scaled does not hold this call's scaled column points");
      } else {
        sim_scale(Cc, nc, nc_pad, p, centre, &C);
        std::memcpy(form.scratch + (size_t)nr_pad * (p.d + 1), C.data(), C.size() * 8);
      }
    }
  }
  for (int tr = 0; tr < nr_pad / 128; ++tr)
    for (int tc = 0; tc < nc_pad / 128; ++tc) {
      if (lower_only && col_g0 + tc * 128 > tr * 128 + 127) continue;
      for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) {
          const int gi = tr * 128 + i, gc = tc * 128 + j;
          double v;
          if (gi < nr && gc < nc) {
            double s = 0.0;
            for (int k = 0; k < p.d; ++k) {
              const double e = R[(long)gi * p.d + k] / p.ell[k] - Cc[(long)gc * p.d + k] / p.ell[k];
              s += e * e;
            }
            v = (mode == 1) ? s : (mode == 2) ? 1.0 : p.sf2 * std::exp(-0.5 * s);
            if (mode == 0 && same_set && gi == col_g0 + gc) v += p.diag_add;
          } else {
            v = (same_set && gi == col_g0 + gc) ? 1.0 : 0.0;
          }
          out[(long)gi * ld + gc] = v;
        }
    }
  return 0;
}

int launch_sum_log_diag(const double* Dinv, int N, double* out, hipStream_t) {
  double acc = 0.0;
  for (int i = 0; i < N; ++i) acc -= std::log(Dinv[(long)(i >> 7) * 128 * 128 + (i & 127) * 128 + (i & 127)]);
  *out = acc;
  return 0;
}
int launch_dot(const double* a, const double* b, int n, double* out, hipStream_t) {
  double acc = 0.0;
  for (int i = 0; i < n; ++i) acc += a[i] * b[i];
  *out = acc;
  return 0;
}
int launch_row_stats(const double* X, long ld, int rows, int n, const double* w, double sf2, double* mean, double* var, hipStream_t) {
  for (int r = 0; r < rows; ++r) {
    double a = 0.0, q = 0.0;
    for (int i = 0; i < n; ++i) { a += X[(long)r * ld + i] * w[i]; q += X[(long)r * ld + i] * X[(long)r * ld + i]; }
    mean[r] = a;
    var[r] = sf2 - q;
  }
  return 0;
}
int launch_gemv(const double* A, long lda, int m, int k, const double* x, double* y, double alpha, double beta, hipStream_t) {
  if (m <= 0) return 0;
  if (k & 1) return gpk_set_err(-1, "gemv: k=%d must be even", k);
  for (int r = 0; r < m; ++r) {
    double acc = 0.0;
    for (int i = 0; i < k; ++i) acc += A[(long)r * lda + i] * x[i];
    y[r] = (beta == 0.0 ? 0.0 : beta * y[r]) + alpha * acc;
  }
  return 0;
}
int launch_gemv_sub(const double* A, long lda, int m, int k, const double* x, double* y, hipStream_t s) {
  return launch_gemv(A, lda, m, k, x, y, -1.0, 1.0, s);
}
int launch_gemvT_sub(const double* A, long lda, int m, int k, const double* x, double* y, double* ws, hipStream_t) {
  if (m <= 0 || k <= 0) return 0;
  for (int c = 0; c < 64 * k; ++c) ws[c] = 0.0;  // the device kernel writes all 64 x k partial sums
  for (int c = 0; c < k; ++c) {
    double acc = 0.0;
    for (int r = 0; r < m; ++r) acc += A[(long)r * lda + c] * x[r];
    y[c] -= acc;
  }
  return 0;
}
int launch_diag_solve(const double* Lp, int w, long ld, const double* Dinv, double* b, int transpose, hipStream_t) {
  (void)Dinv[(long)(w / 128) * 128 * 128 - 1];  // the inverted diagonal tiles of the panel are read
  if (!transpose) {
    for (int i = 0; i < w; ++i) {
      double v = b[i];
      for (int k = 0; k < i; ++k) v -= Lp[(long)i * ld + k] * b[k];
      b[i] = v / Lp[(long)i * ld + i];
    }
  } else {
    for (int i = w - 1; i >= 0; --i) {
      double v = b[i];
      for (int k = i + 1; k < w; ++k) v -= Lp[(long)k * ld + i] * b[k];
      b[i] = v / Lp[(long)i * ld + i];
    }
  }
  return 0;
}
int launch_transpose(const double* in, long ldin, int rows, int cols, double* out, long ldout, hipStream_t) {
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) out[(long)c * ldout + r] = in[(long)r * ldin + c];
  return 0;
}
int launch_fill(double* p, size_t n, double v, hipStream_t) { std::fill(p, p + n, v); return 0; }
int launch_stamp(unsigned long long* out, hipStream_t) { *out = 1; return 0; }
int launch_spin(unsigned long long, hipStream_t) {  // HOSTSIM_SPIN_US: every spin takes that long, so the stream-layout probe sees "time-sliced"
  if (const char* e = std::getenv("HOSTSIM_SPIN_US")) usleep((useconds_t)std::atoi(e));
  return 0;
}
int launch_set_diag(double* p, long ld, int n, double v, hipStream_t) {
  for (int i = 0; i < n; ++i) p[(long)i * ld + i] = v;
  return 0;
}
int launch_unpack_L(const double* Lp, int w, long ld, int h, int c0, double* out, long ldout, hipStream_t) {
  const int nmax = (int)ldout;
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      const int gr = c0 + r, gc = c0 + c;
      if (gr >= nmax || gc >= nmax) continue;
      out[(long)gr * ldout + gc] = (gc <= gr) ? Lp[(long)r * ld + c] : 0.0;
    }
  return 0;
}
int launch_gather2d(const double* K, long ldk, const long long* idx, int n, double* out, hipStream_t) {
  for (int p = 0; p < n; ++p)
    for (int q = 0; q < n; ++q) out[(long)p * n + q] = K[idx[p] * ldk + idx[q]];
  return 0;
}

// ---- preference-GP pieces (k_pref.hip)
static double normcdf_(double z) { return 0.5 * std::erfc(-z * 0.7071067811865476); }
int launch_pref_pair_stats(const long long* uvi, const double* y, const double* f, int npairs, double isq, double* z_out,
                           double* q_out, double* c_out, double* logphi_out, hipStream_t) {
  for (int p = 0; p < npairs; ++p) {
    const double yy = y[p], z = yy * (isq * (f[uvi[2 * p + 1]] - f[uvi[2 * p]]));
    const double phi = normcdf_(z), nz = std::exp(-0.5 * z * z) * 0.3989422804014327, ratio = nz / phi;
    if (z_out) z_out[p] = z;
    if (q_out) q_out[p] = yy * isq * ratio;
    if (c_out) c_out[p] = isq * isq * (z * ratio + ratio * ratio);
    if (logphi_out) logphi_out[p] = std::log(phi);
  }
  return 0;
}
int launch_pref_pair_t(const long long* uvi, const double* f, const double* c, const double* q, int npairs, double* t, hipStream_t) {
  for (int p = 0; p < npairs; ++p) t[p] = c[p] * (f[uvi[2 * p + 1]] - f[uvi[2 * p]]) + q[p];
  return 0;
}
int launch_pref_scatter(const long long* uvi, const double* t, const double* scale, int do_sqrt, int npairs, double* out, hipStream_t) {
  for (int p = 0; p < npairs; ++p) {
    double s = t[p];
    if (scale) s *= do_sqrt ? std::sqrt(scale[p]) : scale[p];
    out[uvi[2 * p + 1]] += s;
    out[uvi[2 * p]] -= s;
  }
  return 0;
}
int launch_pref_gather(const long long* uvi, const double* x, const double* c, int npairs, double* r, hipStream_t) {
  for (int p = 0; p < npairs; ++p) r[p] = std::sqrt(c[p]) * (x[uvi[2 * p + 1]] - x[uvi[2 * p]]);
  return 0;
}
int launch_pref_gather_rows(const long long* uvi, const double* R, long ldr, const double* c, int npairs, int rows, int rows_pad,
                            int npad, double* U, long ldu, hipStream_t) {
  for (int j = 0; j < rows_pad; ++j)
    for (int p = 0; p < npad; ++p)
      U[(long)j * ldu + p] = (j < rows && p < npairs) ? std::sqrt(c[p]) * (R[(long)j * ldr + uvi[2 * p + 1]] - R[(long)j * ldr + uvi[2 * p]]) : 0.0;
  return 0;
}
int launch_pref_build_S(const double* K, long ldk, const long long* uvi, int npairs, double* S, long lds_, hipStream_t) {
  for (int p = 0; p < npairs; ++p)
    for (int q = 0; q < npairs; ++q) {
      const long long up = uvi[2 * p], vp = uvi[2 * p + 1], uq = uvi[2 * q], vq = uvi[2 * q + 1];
      S[(long)p * lds_ + q] = K[vp * ldk + vq] - K[vp * ldk + uq] - K[up * ldk + vq] + K[up * ldk + uq];
    }
  return 0;
}
int launch_pref_build_B(const double* S, long lds_, const double* c, int npairs, int npad, double* B, long ldb, hipStream_t) {
  for (int p = 0; p < npad; ++p)
    for (int q = 0; q < npad; ++q) {
      double v = (p == q) ? 1.0 : 0.0;
      if (p < npairs && q < npairs) v += std::sqrt(c[p]) * S[(long)p * lds_ + q] * std::sqrt(c[q]);
      B[(long)p * ldb + q] = v;
    }
  return 0;
}
int launch_maxabsdiff(const double* a, const double* b, int n, double* out, hipStream_t) {
  double m = 0.0;
  for (int i = 0; i < n; ++i) m = std::fmax(m, std::fabs(a[i] - b[i]));
  *out = m;
  return 0;
}
int launch_sum(const double* a, int n, double* out, hipStream_t) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += a[i];
  *out = s;
  return 0;
}
int launch_sub(const double* a, const double* b, int n, double* out, hipStream_t) {
  for (int i = 0; i < n; ++i) out[i] = a[i] - b[i];
  return 0;
}

// ---- classification pieces (k_class.hip)
int launch_class_stats(const double* y, const double* f, int n, int link, double* g, double* w, double* sw, double* logp, hipStream_t) {
  for (int i = 0; i < n; ++i) {
    const double yy = y[i], ff = f[i], z = yy * ff;
    double gi, wi, lp;
    if (link == 0) {
      const double p = normcdf_(z), r = std::exp(-0.5 * z * z) * 0.3989422804014327 / p;
      lp = std::log(p); gi = yy * r; wi = r * r + z * r;
    } else {
      lp = (z > 0.0) ? -std::log1p(std::exp(-z)) : z - std::log1p(std::exp(z));
      const double pi = 1.0 / (1.0 + std::exp(-ff));
      gi = 0.5 * (yy + 1.0) - pi; wi = pi * (1.0 - pi);
    }
    if (wi < 0.0) wi = 0.0;
    if (g) g[i] = gi;
    if (w) w[i] = wi;
    if (sw) sw[i] = std::sqrt(wi);
    if (logp) logp[i] = lp;
  }
  return 0;
}
int launch_class_ew(const double* p, const double* q, const double* r, int n, int mode, double* out, hipStream_t) {
  for (int i = 0; i < n; ++i) {
    const double pq = p[i] * q[i];
    out[i] = (mode == 0) ? pq + r[i] : (mode == 1) ? r[i] - pq : pq;
  }
  return 0;
}
int launch_class_scale_rows(const double* R, long ldr, const double* sw, int n, int rows, int rows_pad, int npad, double* U, long ldu, hipStream_t) {
  for (int j = 0; j < rows_pad; ++j)
    for (int i = 0; i < npad; ++i) U[(long)j * ldu + i] = (j < rows && i < n) ? R[(long)j * ldr + i] * sw[i] : 0.0;
  return 0;
}
int launch_class_prob(const double* mean, const double* var, int n, int link, double* prob, hipStream_t) {
  for (int i = 0; i < n; ++i) {
    const double v = std::fmax(var[i], 0.0);
    prob[i] = (link == 0) ? normcdf_(mean[i] / std::sqrt(1.0 + v)) : 1.0 / (1.0 + std::exp(-mean[i] / std::sqrt(1.0 + 0.39269908169872414 * v)));
  }
  return 0;
}

int launch_nlml_grad_terms(const double* X, int N, int d, const double* NK, long ldk, const double* alpha, const SEParams& p,
                           double* partial, double* out, hipStream_t) {
  if (d < 1 || d > 32) return gpk_set_err(-1, "nlml gradient: input dimension d=%d outside 1..32", d);
  const int nt = (N + 127) / 128;
  for (long e = 0; e < (long)nt * nt * (d + 2); ++e) partial[e] = 0.0;  // the device kernel writes every tile's partials
  for (int k = 0; k < d + 2; ++k) out[k] = 0.0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j) {
      double r2 = 0.0, dk[32];
      for (int k = 0; k < d; ++k) {
        const double e = X[(long)i * d + k] / p.ell[k] - X[(long)j * d + k] / p.ell[k];
        dk[k] = e * e;
        r2 += dk[k];
      }
      const double kf = p.sf2 * std::exp(-0.5 * r2);
      const double a = -NK[(long)i * ldk + j] - alpha[i] * alpha[j];
      const double akf = (i == j ? 1.0 : 2.0) * a * kf;
      for (int k = 0; k < d; ++k) out[k] += akf * dk[k];
      out[d] += akf;
      if (i == j) out[d + 1] += a;
    }
  return 0;
}
