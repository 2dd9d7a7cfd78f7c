// TEST INFRASTRUCTURE ONLY (tests/hostsim): helpers shared by the drivers -- check macro, a small dense reference
// (SE kernel, Cholesky, substitutions, the regression posterior) and the seeded problem generator.
#pragma once
#include "../../include/gpk.h"
#include "../../include/gpk_testhooks.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static int g_fail = 0;
#define CHECK(cond, ...)                                                       \
  do {                                                                         \
    if (!(cond)) {                                                             \
      std::fprintf(stderr, "FAIL %s:%d  %s  ", __FILE__, __LINE__, #cond);     \
      std::fprintf(stderr, __VA_ARGS__);                                       \
      std::fprintf(stderr, "  [%s]\n", gpk_last_error());                      \
      ++g_fail;                                                                \
    }                                                                          \
  } while (0)

typedef std::vector<double> vec;
static unsigned long long rng_state = 88172645463325252ull;
static double urand() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (double)(rng_state >> 11) / 9007199254740992.0; }

static vec se_kernel(const vec& X, int n, const vec& Z, int m, int d, const vec& lh, bool noise) {
  vec K((size_t)n * m);
  const double sf2 = std::exp(2 * lh[d]), sn2 = std::exp(2 * lh[d + 1]);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      double s = 0;
      for (int k = 0; k < d; ++k) { const double e = (X[i * d + k] - Z[j * d + k]) / std::exp(lh[k]); s += e * e; }
      K[(size_t)i * m + j] = sf2 * std::exp(-0.5 * s) + ((noise && i == j) ? sn2 : 0.0);
    }
  return K;
}
static bool chol(vec& A, int n) {  // in place, lower; upper zeroed
  for (int j = 0; j < n; ++j) {
    double p = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) p -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(p > 0)) return false;
    const double l = std::sqrt(p);
    A[(size_t)j * n + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double v = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = v / l;
    }
    for (int i = 0; i < j; ++i) A[(size_t)i * n + j] = 0.0;
  }
  return true;
}
static void fwd(const vec& L, int n, double* b) { for (int i = 0; i < n; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= L[(size_t)i * n + k] * b[k]; b[i] = v / L[(size_t)i * n + i]; } }
static void bwd(const vec& L, int n, double* b) { for (int i = n - 1; i >= 0; --i) { double v = b[i]; for (int k = i + 1; k < n; ++k) v -= L[(size_t)k * n + i] * b[k]; b[i] = v / L[(size_t)i * n + i]; } }

struct Problem { int N, d, M; vec X, y, Z, lh; };
static Problem make_problem(int N, int d, int M) {
  Problem p{N, d, M, vec((size_t)N * d), vec(N), vec((size_t)M * d), vec(d + 2)};
  for (auto& v : p.X) v = urand();
  for (auto& v : p.Z) v = urand();
  for (int i = 0; i < N; ++i) { double s = 0; for (int k = 0; k < d; ++k) s += (k + 1) * p.X[i * d + k]; p.y[i] = std::sin(s) + 0.1 * (urand() - 0.5); }
  for (int k = 0; k < d; ++k) p.lh[k] = std::log(0.7);
  p.lh[d] = 0.0; p.lh[d + 1] = std::log(0.1);
  return p;
}
struct Posterior { double nlml; vec mean, var; };
static Posterior dense_posterior(const Problem& p, int N) {
  vec Xs(p.X.begin(), p.X.begin() + (size_t)N * p.d);
  vec K = se_kernel(Xs, N, Xs, N, p.d, p.lh, true);
  chol(K, N);
  vec w(p.y.begin(), p.y.begin() + N);
  fwd(K, N, w.data());
  Posterior r{0, vec(p.M), vec(p.M)};
  double q = 0, ld = 0;
  for (int i = 0; i < N; ++i) { q += w[i] * w[i]; ld += std::log(K[(size_t)i * N + i]); }
  r.nlml = 0.5 * q + ld + 0.5 * N * std::log(2 * M_PI);
  vec Kxz = se_kernel(Xs, N, p.Z, p.M, p.d, p.lh, false);
  for (int j = 0; j < p.M; ++j) {
    vec v(N);
    for (int i = 0; i < N; ++i) v[i] = Kxz[(size_t)i * p.M + j];
    fwd(K, N, v.data());
    double a = 0, qq = 0;
    for (int i = 0; i < N; ++i) { a += v[i] * w[i]; qq += v[i] * v[i]; }
    r.mean[j] = a;
    r.var[j] = std::exp(2 * p.lh[p.d]) - qq;
  }
  return r;
}
static double maxdiff(const vec& a, const vec& b) { double m = 0; for (size_t i = 0; i < a.size(); ++i) m = std::fmax(m, std::fabs(a[i] - b[i])); return m; }
static void opt(const char* k, double v) { CHECK(gpk_set_option(k, v) == 0, "option %s", k); }

