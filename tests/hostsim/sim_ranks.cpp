// TEST INFRASTRUCTURE ONLY (tests/hostsim): the REAL-RANK branches of libgpk's host code with 2 and 4 real processes.
// The parent takes a unique id (gpk_comm_unique_id), forks one process per rank, every rank goes
//   gpk_init -> gpk_comm_init(rank, P, uid) -> gpk_fit / gpk_predict / gpk_fit_predict / gpk_bench_posterior (its multi-rank
//   leg) -> a kernel matrix that is not positive definite -> gpk_comm_allreduce_max / gpk_comm_barrier -> gpk_comm_destroy
// through the C-ABI, compiled against the host-memory HIP stand-in and the inter-process RCCL stand-in (sim_rccl.cpp), under
// AddressSanitizer + UBSan.  Every rank checks its results against the dense reference and ships them to the parent, which
// requires all ranks of a job to agree BIT FOR BIT and every job to agree with the one-process job.  A rank that another
// rank leaves alone in a collective fails after HOSTSIM_RCCL_TIMEOUT_S; the parent kills a job that overruns its own limit.
#include "sim_common.h"

#include <csignal>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>

struct Shipped {  // what a rank reports: scenario A (three ways), the not-positive-definite return code, the reduced value
  double nlml[3];
  double mean[3][70], var[3][70];
  int rc_notpd;
  double reduced;
  int fails;
};

// layout: 0 = what gpk_comm_init decides (its stream-layout probe sees the stand-in streams abreast: four streams), 1 = option
// dist_pipeline = 0 (three streams), 2 = the probe sees a time-sliced pair (HOSTSIM_SPIN_US: the stand-in spin takes that long) and
// gives up the substitution stream -- the fallback a real node may take on some boxes and not on others
static int rank_main(int rank, int P, const unsigned char* uid, int out_fd, int layout) {
  Shipped sh{};
  CHECK(gpk_init(0) == 0, "init");
  if (layout == 1) opt("dist_pipeline", 0);
  if (layout == 2) setenv("HOSTSIM_SPIN_US", "40", 1);
  CHECK(gpk_comm_init(rank, P, uid) == 0, "comm init rank %d of %d", rank, P);
  unsetenv("HOSTSIM_SPIN_US");
  CHECK(gpk_comm_rank() == rank && gpk_comm_size() == P, "rank / size");
  if (P > 1) CHECK(gpk_get_option("dist_streams_abreast") == (layout == 0 ? 1 : layout == 1 ? -1 : 0), "stream-layout probe (layout %d): %g, worst %g us",
                   layout, gpk_get_option("dist_streams_abreast"), gpk_get_option("dist_streams_worst_us"));
  const Problem p = make_problem(900, 3, 70);
  const Posterior ref = dense_posterior(p, p.N);
  auto close_to_ref = [&](double nl, const vec& m, const vec& v, const char* what) {
    CHECK(std::fabs(nl - ref.nlml) <= 1e-9 * std::fabs(ref.nlml), "%s nlml %.12g vs %.12g (rank %d of %d)", what, nl, ref.nlml, rank, P);
    CHECK(maxdiff(m, ref.mean) <= 1e-8 && maxdiff(v, ref.var) <= 1e-8, "%s mean/var %g %g (rank %d of %d)", what, maxdiff(m, ref.mean), maxdiff(v, ref.var), rank, P);
  };
  for (int pass = 0; pass < 8; ++pass) {  // default schedule; no look-ahead; three panels per ownership block; panels of three widths; no tail block; tail panels broadcast one by one
    if (pass == 1) opt("lookahead", 0);
    if (pass == 2) { opt("lookahead", 1); opt("dist_group", 3); }
    if (pass == 3) { opt("dist_group", 2); opt("nb_big_until", 256); opt("nb_small_from", 256); }  // 512 / 256 / 128 / 128 at N = 900
    if (pass == 4) { opt("nb_big_until", 8192); opt("nb_small_from", 4096); opt("dist_tail", 0); }
    if (pass == 5) { opt("dist_tail", 1); opt("dist_tail_chunk", 1); }
    if (pass == 6) { opt("dist_tail_chunk", 8); opt("dist_tail_alone", 1); }   // the tail's owner = the last rank, outside the cyclic part (P >= 3)
    if (pass == 7) opt("dist_tail_alone", 0);
    gpk_model* h = nullptr;
    double nl = 0;
    vec mean(p.M), var(p.M);
    CHECK(gpk_fit(p.X.data(), p.y.data(), p.N, p.d, p.lh.data(), 0.0, 0, &h, &nl) == 0, "fit (pass %d)", pass);
    CHECK(gpk_predict(h, p.Z.data(), p.M, mean.data(), var.data()) == 0, "predict (pass %d)", pass);
    close_to_ref(nl, mean, var, "fit + predict");
    if (P > 1) CHECK(gpk_get_option("dist_lookahead_used") == (pass == 1 ? 0 : 1), "look-ahead branch (pass %d)", pass);
    // the last columns as one tail block on one rank (tile pipeline there, chunked broadcasts): 1024 padded rows in two ownership
    // blocks of 512 under the default schedule, the second one is the tail block
    if (P > 1 && pass >= 6) CHECK((gpk_get_option("dist_tail_owner") >= P) == (pass == 6 && P >= 3), "tail owner (pass %d): %g", pass, gpk_get_option("dist_tail_owner"));
    if (P > 1 && (pass == 0 || pass >= 4)) CHECK((gpk_get_option("dist_tail_from") > 0) == (pass != 4), "tail block (pass %d): %g", pass, gpk_get_option("dist_tail_from"));
    if (pass >= 4) CHECK(nl == sh.nlml[0] && std::memcmp(mean.data(), sh.mean[0], sizeof sh.mean[0]) == 0 && std::memcmp(var.data(), sh.var[0], sizeof sh.var[0]) == 0, "tail block on/off, chunked or not: bit for bit (pass %d)", pass);
    if (pass == 0) {
      sh.nlml[0] = nl;
      std::memcpy(sh.mean[0], mean.data(), sizeof sh.mean[0]);
      std::memcpy(sh.var[0], var.data(), sizeof sh.var[0]);
      vec g(p.d + 2), B(p.N, 1.0), Xo(p.N);
      if (P > 1) {  // single-GPU-only entry points refuse a block-cyclic model on every rank alike (no rank enters a collective)
        CHECK(gpk_nlml_grad(h, g.data()) < 0, "gradient on a block-cyclic model");
        CHECK(gpk_solve(h, B.data(), 1, Xo.data()) < 0, "solve on a block-cyclic model");
      }
    }
    gpk_free(h);
  }
  opt("dist_tail_chunk", 8); opt("dist_tail_alone", -1);
  {
    gpk_model* h = nullptr;
    double nl = 0;
    vec mean(p.M), var(p.M);
    CHECK(gpk_fit_predict(p.X.data(), p.y.data(), p.N, p.d, p.lh.data(), 0.0, p.Z.data(), p.M, &h, &nl, mean.data(), var.data()) == 0, "fit_predict");
    close_to_ref(nl, mean, var, "fit_predict");
    sh.nlml[1] = nl;
    std::memcpy(sh.mean[1], mean.data(), sizeof sh.mean[1]);
    std::memcpy(sh.var[1], var.data(), sizeof sh.var[1]);
    gpk_free(h);
    double ms = 0;
    CHECK(gpk_comm_barrier() == 0, "barrier");
    CHECK(gpk_bench_posterior(p.X.data(), p.y.data(), p.N, p.d, p.lh.data(), p.Z.data(), p.M, 1, 2, &ms, &nl, mean.data(), var.data()) == 0, "bench_posterior");
    close_to_ref(nl, mean, var, "bench_posterior");
    sh.nlml[2] = nl;
    std::memcpy(sh.mean[2], mean.data(), sizeof sh.mean[2]);
    std::memcpy(sh.var[2], var.data(), sizeof sh.var[2]);
  }
  // ragged sizes: fewer points than one ownership block, one point, padding-only blocks on some ranks
  for (int N : {1, 130, 300, 513}) {
    Problem q = make_problem(N, 2, 9);
    const Posterior r = dense_posterior(q, N);
    gpk_model* h = nullptr;
    double nl = 0;
    vec mean(q.M), var(q.M);
    CHECK(gpk_fit_predict(q.X.data(), q.y.data(), N, q.d, q.lh.data(), 0.0, q.Z.data(), q.M, &h, &nl, mean.data(), var.data()) == 0, "fit_predict N=%d", N);
    CHECK(std::fabs(nl - r.nlml) <= 1e-9 * std::fabs(r.nlml) + 1e-12 && maxdiff(mean, r.mean) <= 1e-8 && maxdiff(var, r.var) <= 1e-8, "N=%d on %d ranks", N, P);
    gpk_free(h);
  }
  {
    // not positive definite: a NaN coordinate in point 700 makes pivot 701 the first that fails, whatever the schedule.
    // Only the owner of that panel sees it, and the owners of later panels then fail on what is left: every rank must
    // come back with the FIRST failing minor, LAPACK-style (and none may be left waiting in a broadcast).
    vec bad(p.X);
    bad[(size_t)700 * p.d + 1] = NAN;
    gpk_model* h = nullptr;
    sh.rc_notpd = gpk_fit(bad.data(), p.y.data(), p.N, p.d, p.lh.data(), 0.0, 0, &h, nullptr);
    CHECK(sh.rc_notpd == 701 && h == nullptr, "NaN in point 700: rc %d", sh.rc_notpd);
    if (P > 1) {
      vec A(4, 1.0);
      CHECK(gpk_potrf(A.data(), 2) < 0, "gpk_potrf in a multi-rank process");
    }
    // ... and the communicator is still usable afterwards
    double nl = 0;
    CHECK(gpk_fit(p.X.data(), p.y.data(), p.N, p.d, p.lh.data(), 0.0, 0, &h, &nl) == 0 && nl == sh.nlml[0], "fit after the failure");
    gpk_free(h);
  }
  double x[2] = {(double)(10 + rank), -(double)rank};
  CHECK(gpk_comm_allreduce_max(x, 2) == 0 && x[0] == 10.0 + P - 1 && x[1] == 0.0, "allreduce_max: %g %g", x[0], x[1]);
  sh.reduced = x[0];
  CHECK(gpk_comm_barrier() == 0, "barrier");
  CHECK(gpk_comm_destroy() == 0, "comm destroy");
  gpk_shutdown();
  sh.fails = g_fail;
  if (write(out_fd, &sh, sizeof sh) != (ssize_t)sizeof sh) return 3;
  return g_fail ? 1 : 0;
}

// One job of P ranks; results of rank 0 in *first.  Returns false on any failure.
static bool run_job(int P, Shipped* first, double limit_s, int layout = 0) {
  unsigned char uid[GPK_UNIQUE_ID_BYTES];
  if (gpk_comm_unique_id(uid) != 0) { std::fprintf(stderr, "unique id: %s\n", gpk_last_error()); return false; }
  std::vector<pid_t> pid(P);
  std::vector<int> rd(P);
  for (int r = 0; r < P; ++r) {
    int fd[2];
    if (pipe(fd) != 0) return false;
    pid[r] = fork();
    if (pid[r] < 0) return false;
    if (pid[r] == 0) {
      close(fd[0]);
      for (int q = 0; q < r; ++q) close(rd[q]);
      _exit(rank_main(r, P, uid, fd[1], layout));
    }
    close(fd[1]);
    rd[r] = fd[0];
  }
  bool ok = true;
  std::vector<int> done(P, 0);
  int left = P;
  for (double waited = 0; left > 0; waited += 0.05) {
    for (int r = 0; r < P; ++r) {
      if (done[r]) continue;
      int st = 0;
      if (waitpid(pid[r], &st, WNOHANG) == pid[r]) {
        done[r] = 1;
        --left;
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
          std::fprintf(stderr, "job of %d ranks: rank %d ended with status 0x%x\n", P, r, st);
          ok = false;
        }
      }
    }
    if (left > 0 && waited > limit_s) {
      std::fprintf(stderr, "job of %d ranks: still running after %.0f s, killing it\n", P, limit_s);
      for (int r = 0; r < P; ++r) if (!done[r]) { kill(pid[r], SIGKILL); waitpid(pid[r], nullptr, 0); }
      return false;
    }
    if (left > 0) usleep(50000);
  }
  std::vector<Shipped> got(P);
  for (int r = 0; r < P; ++r) {
    if (read(rd[r], &got[r], sizeof(Shipped)) != (ssize_t)sizeof(Shipped)) { std::fprintf(stderr, "job of %d ranks: no report from rank %d\n", P, r); ok = false; }
    close(rd[r]);
  }
  if (!ok) return false;
  for (int r = 1; r < P; ++r) {
    // every rank ends up with the same bits (replicated factor, all-gathered prediction, all-reduced verdict)
    Shipped a = got[0], b = got[r];
    a.fails = b.fails = 0;
    if (std::memcmp(&a, &b, sizeof a) != 0) { std::fprintf(stderr, "job of %d ranks: rank %d disagrees with rank 0\n", P, r); ok = false; }
  }
  *first = got[0];
  return ok;
}

int main() {
  Shipped one{}, many{};
  bool ok = run_job(1, &one, 600.0);
  for (int P : {2, 4}) {
    if (!run_job(P, &many, 600.0)) { ok = false; continue; }
    double worst = 0;
    for (int k = 0; k < 3; ++k) {
      worst = std::fmax(worst, std::fabs(many.nlml[k] - one.nlml[k]) / std::fabs(one.nlml[k]));
      for (int j = 0; j < 70; ++j) worst = std::fmax(worst, std::fmax(std::fabs(many.mean[k][j] - one.mean[k][j]), std::fabs(many.var[k][j] - one.var[k][j])));
    }
    if (!(worst <= 1e-10)) { std::fprintf(stderr, "%d ranks against one process: %g\n", P, worst); ok = false; }
    if (many.rc_notpd != one.rc_notpd) { std::fprintf(stderr, "%d ranks report minor %d, one process %d\n", P, many.rc_notpd, one.rc_notpd); ok = false; }
    std::printf("hostsim ranks: %d processes agree bit for bit with each other and to %.1e with the one-process job\n", P, worst);
    // the three-stream layouts (by option; by the probe's verdict): the same operations in the same order -> the four-stream job's bits
    for (int layout : {1, 2}) {
      if (P == 4 && layout == 1) continue;
      Shipped alt{};
      if (!run_job(P, &alt, 600.0, layout)) { ok = false; continue; }
      Shipped a = many, b = alt;
      a.fails = b.fails = 0;
      if (std::memcmp(&a, &b, sizeof a) != 0) { std::fprintf(stderr, "%d ranks, stream layout %d: results differ from the four-stream job\n", P, layout); ok = false; }
      else std::printf("hostsim ranks: %d processes, %s: bit for bit the four-stream job\n", P, layout == 1 ? "dist_pipeline = 0" : "substitution stream given up after the probe");
    }
  }
  {
    // a rank that never shows up: the others must come back with an error from gpk_comm_init, not hang
    setenv("HOSTSIM_RCCL_TIMEOUT_S", "2", 1);
    unsigned char uid[GPK_UNIQUE_ID_BYTES];
    gpk_comm_unique_id(uid);
    const pid_t c = fork();
    if (c == 0) {
      if (gpk_init(0) != 0) _exit(2);
      const int rc = gpk_comm_init(0, 2, uid);
      gpk_shutdown();
      _exit(rc < 0 ? 0 : 1);
    }
    int st = 0;
    waitpid(c, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { std::fprintf(stderr, "a lone rank did not time out cleanly (status 0x%x)\n", st); ok = false; }
    unsetenv("HOSTSIM_RCCL_TIMEOUT_S");
  }
  if (!ok) { std::fprintf(stderr, "hostsim ranks: FAILED\n"); return 1; }
  std::printf("hostsim ranks: all checks passed\n");
  return 0;
}
