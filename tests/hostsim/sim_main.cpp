// TEST INFRASTRUCTURE ONLY (tests/hostsim): drives the C-ABI of libgpk's HOST code, compiled against the host-memory
// HIP stand-in and the CPU launch_* stand-ins, under AddressSanitizer + UBSan.  Every scenario is also checked
// numerically against a small dense reference written here, so the panel schedules / block-cyclic phases / drivers
// are exercised as real computations, not just for memory safety.  Exit status 0 = all checks passed.
#include "sim_common.h"

static void check_fit_predict(const Problem& p, const char* what) {
  const Posterior ref = dense_posterior(p, p.N);
  gpk_model* h = nullptr;
  double nlml = 0;
  CHECK(gpk_fit(p.X.data(), p.y.data(), p.N, p.d, p.lh.data(), 0.0, 0, &h, &nlml) == 0, "%s fit", what);
  vec mean(p.M), var(p.M);
  CHECK(gpk_predict(h, p.Z.data(), p.M, mean.data(), var.data()) == 0, "%s predict", what);
  CHECK(std::fabs(nlml - ref.nlml) <= 1e-9 * std::fabs(ref.nlml), "%s nlml %.12g vs %.12g", what, nlml, ref.nlml);
  CHECK(maxdiff(mean, ref.mean) <= 1e-8 && maxdiff(var, ref.var) <= 1e-8, "%s mean/var %g %g", what, maxdiff(mean, ref.mean), maxdiff(var, ref.var));
  gpk_free(h);
  h = nullptr;
  vec m2(p.M), v2(p.M);
  double nl2 = 0;
  CHECK(gpk_fit_predict(p.X.data(), p.y.data(), p.N, p.d, p.lh.data(), 0.0, p.Z.data(), p.M, &h, &nl2, m2.data(), v2.data()) == 0, "%s fused", what);
  CHECK(std::fabs(nl2 - ref.nlml) <= 1e-9 * std::fabs(ref.nlml) && maxdiff(m2, ref.mean) <= 1e-8 && maxdiff(v2, ref.var) <= 1e-8, "%s fused result", what);
  vec g(p.d + 2);
  CHECK(gpk_nlml_grad(h, g.data()) == 0, "%s grad", what);
  for (double v : g) CHECK(std::isfinite(v), "%s grad finite", what);
  gpk_free(h);
}

int main() {
  CHECK(gpk_abi_version() == GPK_ABI_VERSION, "abi");
  CHECK(gpk_fit(nullptr, nullptr, 1, 1, nullptr, 0, 0, nullptr, nullptr) < 0, "null pointers are refused");
  CHECK(gpk_init(0) == 0, "init");
  CHECK(gpk_comm_init(0, 1, nullptr) == 0, "one-rank comm");

  // ---- potrf on caller matrices, ragged sizes, not positive definite
  for (int N : {1, 2, 127, 128, 129, 300, 700}) {
    vec A((size_t)N * N);
    for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) { const double v = (i == j ? N + 1.0 : 0.0) + 1.0 / (1.0 + i - j); A[(size_t)i * N + j] = A[(size_t)j * N + i] = v; }
    vec L = A, R = A;
    CHECK(gpk_potrf(L.data(), N) == 0, "potrf N=%d", N);
    chol(R, N);
    CHECK(maxdiff(L, R) <= 1e-10, "potrf N=%d diff %g", N, maxdiff(L, R));
  }
  {
    const int N = 300;
    vec A((size_t)N * N, 0.0);
    for (int i = 0; i < N; ++i) A[(size_t)i * N + i] = 2.0;
    A[(size_t)131 * N + 131] = -1.0;
    CHECK(gpk_potrf(A.data(), N) == 132, "LAPACK-style info");
  }

  // ---- the posterior under every panel schedule the host code knows
  check_fit_predict(make_problem(300, 3, 40), "default");
  check_fit_predict(make_problem(1, 1, 3), "N=1");
  opt("leaf_fat_rows", -1); check_fit_predict(make_problem(800, 3, 40), "slot-sized leaf everywhere");
  opt("leaf_fat_rows", 6144);
  opt("nb", 128); check_fit_predict(make_problem(900, 2, 33), "nb=128 (paired updates)");
  opt("pair_update", 0); check_fit_predict(make_problem(700, 2, 33), "nb=128 one panel per update");
  opt("pair_update", 1); opt("lookahead", 0); check_fit_predict(make_problem(500, 4, 20), "no look-ahead");
  opt("lookahead", 1); opt("fuse_predict", 0); check_fit_predict(make_problem(500, 4, 20), "unfused prediction");
  opt("fuse_predict", 1); opt("nb", 0);
  // far-column updates of the pipelined prediction deferred to a boundary (default schedule, two boundaries; and two-level below)
  opt("pred_defer_from", 512); check_fit_predict(make_problem(1500, 3, 40), "deferred far updates from column 512");
  opt("pred_defer_from", 1152); check_fit_predict(make_problem(1500, 3, 40), "deferred far updates from column 1152 (the next panel start is taken)");
  // two-level blocking forced onto small problems (active from Npad >= 3 * outer_until)
  opt("nb_outer", 1024); opt("outer_until", 256); opt("nb_big_until", 512); opt("nb_small_from", 256); opt("tail_rows", 0);
  opt("pred_defer_from", 1024);
  check_fit_predict(make_problem(1300, 3, 50), "two-level blocking, far updates deferred to an outer boundary");
  opt("pred_defer_from", 0);
  check_fit_predict(make_problem(1300, 3, 50), "two-level blocking");
  opt("outer_until", 0); opt("nb_outer", 768); check_fit_predict(make_problem(1100, 2, 10), "two-level blocking, 768");
  opt("nb_outer", 0); opt("nb_small_from", 640); opt("tail_rows", 640); opt("nb_big_until", 512);
  check_fit_predict(make_problem(1400, 2, 30), "tile pipeline after a hand-over");
  CHECK(gpk_get_option("potrf_tail_from") > 0, "tile pipeline used after the hand-over");
  opt("nb", 128); opt("tail_rows", 1 << 30);
  check_fit_predict(make_problem(900, 3, 20), "tile pipeline for the whole factorisation");
  CHECK(gpk_get_option("potrf_tail_from") == 0, "tile pipeline from panel 0");
  for (int cf : {1, 2}) {  // one launch per chain step (flags set / polled across streams; the stand-ins check the producer came first)
    opt("chain_fused", cf);
    check_fit_predict(make_problem(900, 3, 20), "tile pipeline with the fused chain step");
    CHECK(gpk_get_option("potrf_chain_fused") == 7, "fused chain steps: %g", gpk_get_option("potrf_chain_fused"));
  }
  opt("chain_fused", 0);
  check_fit_predict(make_problem(300, 1, 5), "tile pipeline, three tiles");
  opt("tail_rows", 0); opt("nb", 0);
  opt("nb_outer", 2048); opt("outer_until", 8192); opt("nb_big_until", 8192); opt("nb_small_from", 2048);

  // ---- lazy corner: trapezoid updates before the boundary, catch-up launches over K-segment tables on the corner stream,
  //      the boundary panel's update merged with the last deferred panels; with and without a tile-pipeline tail behind it.
  //      The stand-ins apply a table segment by segment, so the factor must equal the immediate schedule's exactly.
  {
    opt("pair_update", 0);
    auto spd = [](int N) {
      vec A((size_t)N * N);
      for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) { const double v = (i == j ? N + 1.0 : 0.0) + std::cos(0.07 * (i - j)); A[(size_t)i * N + j] = A[(size_t)j * N + i] = v; }
      return A;
    };
    {
      const int N = 900;  // eight 128-wide panels, the last four through the tile pipeline or not
      opt("nb", 128);
      const vec A = spd(N);
      for (int tail : {0, 512}) {
        opt("tail_rows", tail);
        opt("corner_from", 0);
        vec L0 = A;
        CHECK(gpk_potrf(L0.data(), N) == 0 && gpk_get_option("potrf_corner_from") == 0, "immediate schedule");
        struct { int from, chunk, last, expect, expect_tail; } cases[] = {{512, 256, 128, 512, 512}, {384, 128, 256, 384, 384}, {256, 4096, 0, 256, 256},
                                                                           {640, 256, 128, 640, 0}, {128, 256, 128, 256, 256}, {896, 256, 128, 0, 0}};
        for (auto& c : cases) {
          opt("corner_from", c.from); opt("corner_chunk", c.chunk); opt("corner_last", c.last);
          vec L1 = A;
          const int want = tail ? c.expect_tail : c.expect;
          CHECK(gpk_potrf(L1.data(), N) == 0, "lazy corner from %d", c.from);
          CHECK(gpk_get_option("potrf_corner_from") == want, "lazy corner from %d: boundary %g, expected %d (tail %d)", c.from, gpk_get_option("potrf_corner_from"), want, tail);
          CHECK(maxdiff(L0, L1) == 0.0, "lazy corner from %d chunk %d last %d tail %d: diff %g", c.from, c.chunk, c.last, tail, maxdiff(L0, L1));
        }
      }
      // nested regions, held-back catch-up
      opt("tail_rows", 0);
      vec L0 = A;
      opt("corner_from", 0);
      CHECK(gpk_potrf(L0.data(), N) == 0, "immediate schedule");
      struct { int b1, b2, b3, hold; } nest[] = {{256, 512, 768, 0}, {384, 640, 0, 256}, {256, 384, 512, 1 << 20}};
      for (auto& c : nest) {
        opt("corner_from", c.b1); opt("corner_from2", c.b2); opt("corner_from3", c.b3); opt("corner_hold", c.hold); opt("corner_chunk", 256); opt("corner_last", 128);
        opt("corner_grid", c.hold ? 0 : 256);
        vec L1 = A;
        CHECK(gpk_potrf(L1.data(), N) == 0, "nested lazy regions %d %d %d", c.b1, c.b2, c.b3);
        CHECK(gpk_get_option("potrf_corner_from") == c.b1 && gpk_get_option("potrf_corner_from2") == c.b2 && gpk_get_option("potrf_corner_from3") == c.b3, "nested boundaries");
        CHECK(maxdiff(L0, L1) == 0.0, "nested lazy regions %d %d %d hold %d: diff %g", c.b1, c.b2, c.b3, c.hold, maxdiff(L0, L1));
      }
      opt("corner_from2", 0); opt("corner_from3", 0); opt("corner_hold", 0); opt("corner_grid", 256);
      opt("corner_from", 512); opt("corner_chunk", 256); opt("corner_last", 128);
      vec B((size_t)N * N, 0.0);
      for (int i = 0; i < N; ++i) B[(size_t)i * N + i] = 2.0;
      B[(size_t)700 * N + 700] = -1.0;
      CHECK(gpk_potrf(B.data(), N) == 701, "not positive definite inside the corner");
    }
    // panels of three widths (segments with different leading dimensions in one table), posterior behind the corner
    opt("nb", 0); opt("nb_big_until", 512); opt("nb_small_from", 384); opt("tail_rows", 0);
    opt("corner_from", 768); opt("corner_chunk", 4096); opt("corner_last", 512);
    check_fit_predict(make_problem(1300, 3, 30), "posterior behind a lazy corner (512 / 256 / 128 panels)");
    CHECK(gpk_get_option("potrf_corner_from") == 768, "boundary of the three-width schedule: %g", gpk_get_option("potrf_corner_from"));
    opt("corner_from", 0); opt("corner_chunk", 2048); opt("corner_last", 256);
    opt("nb_big_until", 8192); opt("nb_small_from", 2048); opt("tail_rows", 0); opt("pair_update", 1);
  }

  {
    // K in two launches (block column 0, then the triangle behind it: rows_ready / rows_norms of the scalar-operand form), panel 0
    // factored behind the first one
    opt("nb", 128); opt("tail_rows", 512); opt("kbuild_split", 1);
    check_fit_predict(make_problem(1300, 3, 30), "posterior with the two-launch K build");
    CHECK(gpk_get_option("kbuild_split_used") == 1, "two-launch K build used");
    opt("se_scalar", 0);
    check_fit_predict(make_problem(1300, 3, 30), "posterior with the two-launch K build, LDS form");
    CHECK(gpk_get_option("kbuild_split_used") == 1, "two-launch K build used (LDS form)");
    opt("kbuild_split", 0);
    check_fit_predict(make_problem(1300, 3, 30), "posterior with the one-launch K build, LDS form");
    CHECK(gpk_get_option("kbuild_split_used") == 0, "two-launch K build off");
    opt("se_scalar", 1); opt("nb", 0); opt("tail_rows", 0);
  }

  // ---- block-cyclic schedule on virtual ranks (look-ahead and plain), against the one-rank factor
  for (int la : {1, 0}) for (int grp : {1, 2, 3}) {
    opt("lookahead", la);
    opt("dist_group", grp);
    opt("nb", 128);
    const int N = 700;
    vec A((size_t)N * N);
    for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) { const double v = (i == j ? N + 1.0 : 0.0) + std::cos(0.1 * (i - j)); A[(size_t)i * N + j] = A[(size_t)j * N + i] = v; }
    vec L1((size_t)N * N), LP((size_t)N * N);
    CHECK(gpk_test_dist_potrf(A.data(), N, 1, L1.data()) == 0, "dist P=1");
    for (int P : {2, 3, 4, 8}) {
      CHECK(gpk_test_dist_potrf(A.data(), N, P, LP.data()) == 0, "dist P=%d la=%d", P, la);
      CHECK(gpk_get_option("dist_lookahead_used") == la, "look-ahead branch P=%d", P);
      CHECK(maxdiff(L1, LP) <= 1e-11, "dist P=%d la=%d group=%d diff %g", P, la, grp, maxdiff(L1, LP));
    }
  }
  opt("lookahead", 1);
  // ---- the tail block: the last tail_rows columns live on one rank, which runs the tile pipeline over them and broadcasts its panels
  for (int grp : {1, 2, 3}) for (int fused : {0, 2}) {
    opt("dist_group", grp); opt("nb", 128); opt("tail_rows", 768); opt("chain_fused", fused); opt("dist_tail_chunk", fused ? 4 : 1);
    opt("dist_tail_alone", fused ? 0 : -1);  // (-1: the work model decides whether the tail's owner keeps out of the cyclic part -- here it does from P = 3 on)
    const int N = 1700;
    vec A((size_t)N * N);
    for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) { const double v = (i == j ? N + 1.0 : 0.0) + std::cos(0.1 * (i - j)); A[(size_t)i * N + j] = A[(size_t)j * N + i] = v; }
    vec L1((size_t)N * N), LP((size_t)N * N), L0((size_t)N * N);
    CHECK(gpk_test_dist_potrf(A.data(), N, 1, L1.data()) == 0, "tail block P=1");
    for (int P : {2, 3, 4, 8}) {
      opt("dist_tail", 1);
      CHECK(gpk_test_dist_potrf(A.data(), N, P, LP.data()) == 0, "tail block P=%d", P);
      CHECK(gpk_get_option("dist_tail_from") > 0, "tail block used P=%d group=%d: %g", P, grp, gpk_get_option("dist_tail_from"));
      CHECK((gpk_get_option("dist_tail_owner") >= P) == (!fused && P >= 3), "tail owner outside the cyclic part P=%d fused=%d: %g", P, fused, gpk_get_option("dist_tail_owner"));
      CHECK(maxdiff(L1, LP) == 0.0, "tail block P=%d group=%d diff %g", P, grp, maxdiff(L1, LP));
      opt("dist_tail", 0);
      CHECK(gpk_test_dist_potrf(A.data(), N, P, L0.data()) == 0, "no tail block P=%d", P);
      CHECK(gpk_get_option("dist_tail_from") < 0, "tail block off P=%d", P);
      CHECK(maxdiff(L0, LP) == 0.0, "tail block on/off P=%d group=%d diff %g", P, grp, maxdiff(L0, LP));
    }
    opt("dist_tail", 1); opt("chain_fused", 0); opt("dist_tail_chunk", 8); opt("dist_tail_alone", -1);
  }
  {
    Problem p = make_problem(1500, 3, 70);
    const Posterior ref = dense_posterior(p, p.N);
    opt("dist_group", 1); opt("nb", 128); opt("tail_rows", 640);
    for (int P : {2, 3, 5}) {
      vec nl(P), mean((size_t)P * p.M), var((size_t)P * p.M);
      CHECK(gpk_test_dist_posterior(p.X.data(), p.y.data(), p.N, p.d, p.lh.data(), p.Z.data(), p.M, P, nl.data(), mean.data(), var.data()) == 0, "tail-block posterior P=%d", P);
      CHECK(gpk_get_option("dist_tail_from") > 0, "tail block used in the posterior P=%d", P);
      for (int r = 0; r < P; ++r) {
        CHECK(std::fabs(nl[r] - ref.nlml) <= 1e-9 * std::fabs(ref.nlml), "tail-block posterior nlml rank %d", r);
        CHECK(maxdiff(vec(mean.begin() + (size_t)r * p.M, mean.begin() + (size_t)(r + 1) * p.M), ref.mean) <= 1e-8, "tail-block mean rank %d", r);
        CHECK(maxdiff(vec(var.begin() + (size_t)r * p.M, var.begin() + (size_t)(r + 1) * p.M), ref.var) <= 1e-8, "tail-block var rank %d", r);
      }
    }
  }
  opt("dist_group", 2);
  {
    // the automatic width schedule on block-cyclic models (512 / 256 / 128 by remaining size, thresholds scaled down):
    // ownership blocks of 512 columns, panels of three widths inside them
    opt("nb", 0); opt("nb_big_until", 512); opt("nb_small_from", 384); opt("tail_rows", 0);
    const int N = 1700;
    vec A((size_t)N * N);
    for (int i = 0; i < N; ++i) for (int j = 0; j <= i; ++j) { const double v = (i == j ? N + 1.0 : 0.0) + std::cos(0.1 * (i - j)); A[(size_t)i * N + j] = A[(size_t)j * N + i] = v; }
    vec L1((size_t)N * N), LP((size_t)N * N);
    CHECK(gpk_test_dist_potrf(A.data(), N, 1, L1.data()) == 0, "auto widths P=1");
    for (int P : {2, 3, 4}) {
      CHECK(gpk_test_dist_potrf(A.data(), N, P, LP.data()) == 0, "auto widths P=%d", P);
      CHECK(gpk_get_option("dist_lookahead_used") == 1, "look-ahead branch, auto widths P=%d", P);
      CHECK(maxdiff(L1, LP) <= 1e-11, "auto widths P=%d diff %g", P, maxdiff(L1, LP));
    }
    opt("nb_big_until", 8192); opt("nb_small_from", 2048); opt("nb", 128);
  }
  {
    Problem p = make_problem(900, 3, 70);
    const Posterior ref = dense_posterior(p, p.N);
    for (int P : {2, 4, 8}) {
      vec nl(P), mean((size_t)P * p.M), var((size_t)P * p.M);
      CHECK(gpk_test_dist_posterior(p.X.data(), p.y.data(), p.N, p.d, p.lh.data(), p.Z.data(), p.M, P, nl.data(), mean.data(), var.data()) == 0, "dist posterior P=%d", P);
      for (int r = 0; r < P; ++r) {
        CHECK(std::fabs(nl[r] - ref.nlml) <= 1e-9 * std::fabs(ref.nlml), "dist posterior nlml rank %d", r);
        CHECK(maxdiff(vec(mean.begin() + (size_t)r * p.M, mean.begin() + (size_t)(r + 1) * p.M), ref.mean) <= 1e-8, "dist mean rank %d", r);
        CHECK(maxdiff(vec(var.begin() + (size_t)r * p.M, var.begin() + (size_t)(r + 1) * p.M), ref.var) <= 1e-8, "dist var rank %d", r);
      }
    }
    vec L((size_t)p.N * p.N);
    CHECK(gpk_test_dist_fit(p.X.data(), p.N, p.d, p.lh.data(), 0.0, 4, L.data()) == 0, "dist fit");
  }
  opt("nb", 0); opt("nb_small_from", 4096); opt("tail_rows", 4096);

  // ---- solves, K products, gathers on one handle
  {
    Problem p = make_problem(333, 2, 5);
    gpk_model* h = nullptr;
    CHECK(gpk_fit(p.X.data(), p.y.data(), p.N, p.d, p.lh.data(), 0.0, 1, &h, nullptr) == 0, "fit keep-K");
    vec K = se_kernel(p.X, p.N, p.X, p.N, p.d, p.lh, true), L = K;
    chol(L, p.N);
    for (int nrhs : {3, 20}) {
      vec B((size_t)p.N * nrhs), Xo(B.size()), Lo(B.size()), KB(B.size());
      for (auto& v : B) v = urand() - 0.5;
      CHECK(gpk_solve(h, B.data(), nrhs, Xo.data()) == 0, "solve nrhs=%d", nrhs);
      CHECK(gpk_solve_lower(h, B.data(), nrhs, Lo.data()) == 0, "solve_lower nrhs=%d", nrhs);
      CHECK(gpk_kmatvec(h, B.data(), nrhs, KB.data()) == 0, "kmatvec");
      for (int j = 0; j < nrhs; ++j) {
        vec c(p.N), l(p.N), kb(p.N, 0.0);
        for (int i = 0; i < p.N; ++i) c[i] = l[i] = B[(size_t)i * nrhs + j];
        fwd(L, p.N, l.data());
        fwd(L, p.N, c.data()); bwd(L, p.N, c.data());
        for (int i = 0; i < p.N; ++i) for (int k = 0; k < p.N; ++k) kb[i] += K[(size_t)i * p.N + k] * B[(size_t)k * nrhs + j];
        double e1 = 0, e2 = 0, e3 = 0;
        for (int i = 0; i < p.N; ++i) { e1 = std::fmax(e1, std::fabs(c[i] - Xo[(size_t)i * nrhs + j])); e2 = std::fmax(e2, std::fabs(l[i] - Lo[(size_t)i * nrhs + j])); e3 = std::fmax(e3, std::fabs(kb[i] - KB[(size_t)i * nrhs + j])); }
        CHECK(e1 <= 1e-7 && e2 <= 1e-9 && e3 <= 1e-10, "solve nrhs=%d col %d: %g %g %g", nrhs, j, e1, e2, e3);
      }
    }
    vec Lg((size_t)p.N * p.N);
    CHECK(gpk_model_get_L(h, Lg.data()) == 0 && maxdiff(Lg, L) <= 1e-10, "get_L");
    long long idx[3] = {5, 332, 0};
    double G9[9];
    CHECK(gpk_kgather(h, idx, 3, G9) == 0 && std::fabs(G9[1] - K[(size_t)5 * p.N + 332]) <= 1e-14, "kgather");
    long long bad[1] = {333};
    CHECK(gpk_kgather(h, bad, 1, G9) < 0, "kgather range check");
    gpk_free(h);
  }

  // ---- preference Laplace loop + latent prediction (runs, converges, finite)
  {
    const int nx = 60, np = 30, d = 2;
    Problem p = make_problem(nx, d, 7);
    vec lh = {std::log(0.3), std::log(0.3), 0.0, -INFINITY};
    gpk_model* h = nullptr;
    CHECK(gpk_fit(p.X.data(), nullptr, nx, d, lh.data(), 1e-6, 1, &h, nullptr) == 0, "pref fit");
    std::vector<long long> uvi(2 * np);
    vec y(np), f(nx, 0.0), trace(50);
    for (int i = 0; i < np; ++i) { uvi[2 * i] = 2 * i; uvi[2 * i + 1] = 2 * i + 1; y[i] = (p.X[(2 * i) * d] < p.X[(2 * i + 1) * d]) ? 1.0 : -1.0; }
    double lml = 0; int it = 0;
    CHECK(gpk_pref_laplace(h, uvi.data(), y.data(), np, 1.0, f.data(), 1e-8, 50, &lml, &it, trace.data()) == 0, "pref laplace");
    CHECK(it > 1 && it < 50 && std::isfinite(lml) && trace[it - 1] <= 1e-8, "pref convergence it=%d", it);
    vec mean(7), var(7);
    CHECK(gpk_pref_predict(h, uvi.data(), y.data(), np, 1.0, f.data(), p.Z.data(), 7, mean.data(), var.data()) == 0, "pref predict");
    for (int i = 0; i < 7; ++i) CHECK(std::isfinite(mean[i]) && var[i] > -1e-9 && var[i] <= 1.0 + 1e-9, "pref predict values");
    uvi[3] = nx;
    CHECK(gpk_pref_laplace(h, uvi.data(), y.data(), np, 1.0, f.data(), 1e-8, 50, &lml, &it, nullptr) < 0, "pair index range check");
    gpk_free(h);
  }

  // ---- classification Laplace loop: mode = K grad, both links
  for (int link : {0, 1}) {
    const int n = 150, d = 2;
    Problem p = make_problem(n, d, 9);
    vec lh = {std::log(0.3), std::log(0.3), std::log(1.5), -INFINITY};
    gpk_model* h = nullptr;
    CHECK(gpk_fit(p.X.data(), nullptr, n, d, lh.data(), 0.0, 3, &h, nullptr) == 0, "K-only fit");
    vec y(n), f(n, 0.0), tr(60);
    for (int i = 0; i < n; ++i) y[i] = (p.y[i] > 0.3) ? 1.0 : -1.0;
    double lml = 0; int it = 0;
    CHECK(gpk_class_laplace(h, y.data(), link, f.data(), 1e-9, 60, &lml, &it, tr.data()) == 0, "class laplace link %d", link);
    CHECK(it > 1 && it < 60 && std::isfinite(lml), "class convergence it=%d", it);
    vec K = se_kernel(p.X, n, p.X, n, d, lh, false);
    double worst = 0;
    for (int i = 0; i < n; ++i) {
      double kf = 0;
      for (int j = 0; j < n; ++j) {
        const double z = y[j] * f[j];
        const double g = (link == 0) ? y[j] * std::exp(-0.5 * z * z) * 0.3989422804014327 / (0.5 * std::erfc(-z * 0.7071067811865476))
                                     : 0.5 * (y[j] + 1.0) - 1.0 / (1.0 + std::exp(-f[j]));
        kf += K[(size_t)i * n + j] * g;
      }
      worst = std::fmax(worst, std::fabs(kf - f[i]));
    }
    CHECK(worst <= 1e-7, "class fixed point f = K grad: %g", worst);
    vec mean(9), var(9), prob(9);
    CHECK(gpk_class_predict(h, y.data(), link, f.data(), p.Z.data(), 9, mean.data(), var.data(), prob.data()) == 0, "class predict");
    for (int i = 0; i < 9; ++i) CHECK(prob[i] > 0 && prob[i] < 1 && var[i] > -1e-9, "class predict values");
    vec dummy(n);
    CHECK(gpk_solve(h, y.data(), 1, dummy.data()) < 0, "K-only handle has no factor");
    y[4] = 0.5;
    CHECK(gpk_class_laplace(h, y.data(), link, f.data(), 1e-9, 60, &lml, &it, nullptr) < 0, "labels are checked");
    gpk_free(h);
  }

  // ---- growing model: append across tile boundaries, in place and re-allocated, against fresh fits
  {
    Problem p = make_problem(600, 2, 11);
    gpk_model* h = nullptr;
    double nl = 0;
    int n = 100;
    CHECK(gpk_fit(p.X.data(), p.y.data(), n, p.d, p.lh.data(), 0.0, 4, &h, &nl) == 0, "growable fit");
    for (int k : {1, 27, 1, 200, 130, 141}) {
      CHECK(gpk_append(&h, p.X.data() + (size_t)n * p.d, p.y.data() + n, k, &nl) == 0, "append %d at %d", k, n);
      n += k;
      const Posterior ref = dense_posterior(p, n);
      vec mean(p.M), var(p.M);
      CHECK(gpk_predict(h, p.Z.data(), p.M, mean.data(), var.data()) == 0, "predict after append");
      CHECK(std::fabs(nl - ref.nlml) <= 1e-9 * std::fabs(ref.nlml) && maxdiff(mean, ref.mean) <= 1e-8 && maxdiff(var, ref.var) <= 1e-8,
            "append to n=%d: nlml %.12g vs %.12g, mean %g var %g", n, nl, ref.nlml, maxdiff(mean, ref.mean), maxdiff(var, ref.var));
    }
    gpk_free(h);
    gpk_model* plain = nullptr;
    CHECK(gpk_fit(p.X.data(), p.y.data(), 200, p.d, p.lh.data(), 0.0, 0, &plain, nullptr) == 0, "plain fit");
    CHECK(gpk_append(&plain, p.X.data(), p.y.data(), 1, nullptr) < 0, "append needs a growable model");
    gpk_free(plain);
  }

  // ---- kernel-matrix entry points, error paths, shutdown / re-init with a warm pool
  {
    Problem p = make_problem(77, 3, 13);
    vec K((size_t)77 * 77), Kz((size_t)77 * 13), D((size_t)77 * 13);
    CHECK(gpk_se_kxx(p.X.data(), 77, 3, p.lh.data(), 0.0, K.data()) == 0 && maxdiff(K, se_kernel(p.X, 77, p.X, 77, 3, p.lh, true)) <= 1e-13, "kxx");
    CHECK(gpk_se_kxz(p.X.data(), 77, p.Z.data(), 13, 3, p.lh.data(), Kz.data()) == 0 && maxdiff(Kz, se_kernel(p.X, 77, p.Z, 13, 3, p.lh, false)) <= 1e-13, "kxz");
    CHECK(gpk_sqdist(p.X.data(), 77, p.Z.data(), 13, 3, D.data()) == 0, "sqdist");
    CHECK(gpk_se_kxx(p.X.data(), 77, 33, p.lh.data(), 0.0, K.data()) < 0, "d range check");
    CHECK(gpk_set_option("no such option", 1.0) < 0, "unknown option");
    vec lh0 = p.lh; lh0[4] = -INFINITY;
    vec dup(p.X); for (int k = 0; k < 3; ++k) dup[3 + k] = dup[k];
    gpk_model* h = nullptr;
    CHECK(gpk_fit(dup.data(), p.y.data(), 77, 3, lh0.data(), 0.0, 0, &h, nullptr) > 0 && h == nullptr, "singular K -> info > 0, no handle");
    CHECK(gpk_test_rccl_selfcheck(4096) == 0, "rccl self check");
  }
  gpk_comm_destroy();
  gpk_shutdown();
  CHECK(gpk_init(0) == 0, "re-init after shutdown");
  check_fit_predict(make_problem(260, 2, 5), "after re-init");
  gpk_shutdown();
  if (g_fail) { std::fprintf(stderr, "%d check(s) failed\n", g_fail); return 1; }
  std::printf("hostsim: all checks passed\n");
  return 0;
}
