// Runtime + blocked Cholesky driver + fused fit/predict (host side of libgpk.so).
#include "gpk_core.h"
#include <memory>
#include <cmath>
#include <chrono>

Runtime G;
int g_potrf_lookahead_used = 0;
int g_potrf_tail_from = -1;  // first panel the last factorisation sent through the tile pipeline (-1: none)
std::vector<int> g_last_pc0, g_last_pw;  // panel layout of the last single-GPU factorisation (gpk_get_option("last_panel_c0_<p>"): bench.py's phase split)
int g_potrf_chain_fused = 0;  // steps of the last tile pipeline that ran as one launch (potf2_chain_kernel)
extern "C" int gpk_comm_allreduce_max(double* inout, int n);
static thread_local std::string g_err = "";

int gpk_set_err(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
const char* gpk_err_cstr() { return g_err.c_str(); }

int rt_require() {
  if (!G.inited) return gpk_set_err(-1, "gpk_init() has not been called");
  return 0;
}

int launch_gemm_auto(const GemmNT& g, hipStream_t s) {
  if (g.cyc_P == 0 && g.m % 128 == 0 && g.n % 128 == 0 && (long)(g.m / 128) * (g.n / 128) >= G.xcd_sq_min_tiles) {
    GemmNT h = g;
    h.xcd_sq = 1;
    return launch_gemm_nt(h, true, s);
  }
  return launch_gemm_nt(g, true, s);
}

// ------------------------------------------------------------------ allocator
int dev_alloc(void** p, size_t bytes) {
  if (bytes == 0) bytes = 256;
  bytes = (bytes + 255) & ~(size_t)255;
  auto it = G.pool.find(bytes);
  if (it != G.pool.end() && !it->second.empty()) {
    *p = it->second.back();
    it->second.pop_back();
    G.pool_bytes -= bytes;
    return 0;
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    // drop the cache and retry once
    for (auto& kv : G.pool)
      for (void* q : kv.second) (void)hipFree(q);
    G.pool.clear();
    G.pool_bytes = 0;
    e = hipMalloc(p, bytes);
    if (e != hipSuccess) return gpk_set_err(-2, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  return 0;
}
void dev_free(void* p, size_t bytes) {
  if (!p) return;
  if (bytes == 0) bytes = 256;
  bytes = (bytes + 255) & ~(size_t)255;
  G.pool[bytes].push_back(p);
  G.pool_bytes += bytes;
}

// ------------------------------------------------------------------ timers
static hipEvent_t ev_get() {
  if (!G.ev_free.empty()) {
    hipEvent_t e = G.ev_free.back();
    G.ev_free.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
void timer_begin(const char* stage, hipStream_t s, double work) {
  if (!G.timing) return;
  StageTimer& t = G.timers[stage];
  t.work_pending += work;
  StageTimer::Rec r;
  r.a = ev_get();
  r.b = nullptr;
  (void)hipEventRecord(r.a, s);
  t.recs.push_back(r);
}
void timer_end(const char* stage, hipStream_t s) {
  if (!G.timing) return;
  StageTimer& t = G.timers[stage];
  if (t.recs.empty()) return;
  StageTimer::Rec& r = t.recs.back();
  r.b = ev_get();
  (void)hipEventRecord(r.b, s);
}
void timers_reset() {
  for (auto& kv : G.timers) {
    for (auto& r : kv.second.recs) {
      if (r.a) G.ev_free.push_back(r.a);
      if (r.b) G.ev_free.push_back(r.b);
    }
    kv.second.recs.clear();
    kv.second.ms = 0;
    kv.second.count = 0;
    kv.second.work = 0;
    kv.second.work_pending = 0;
  }
}
int timers_resolve() {
  GPK_HIP(hipDeviceSynchronize());
  for (auto& kv : G.timers) {
    StageTimer& t = kv.second;
    for (auto& r : t.recs) {
      if (r.a && r.b) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
          t.ms += ms;
          t.count += 1;
        }
      }
      if (r.a) G.ev_free.push_back(r.a);
      if (r.b) G.ev_free.push_back(r.b);
    }
    t.recs.clear();
    t.work += t.work_pending;
    t.work_pending = 0;
  }
  return 0;
}

// ------------------------------------------------------------------ model
int model_create(gpk_model** out, int N, int d, bool keep_k, int P, int r, bool growable) {
  GPK_TRY(rt_require());
  if (N < 1) return gpk_set_err(-1, "N=%d must be >= 1", N);
  if (d < 1 || d > 32) return gpk_set_err(-1, "input dimension d=%d outside 1..32", d);
  gpk_model* m = new gpk_model();
  m->N = N;
  m->d = d;
  int nb = G.nb_auto() ? G.nb_mid : G.nb / 128 * 128;
  if (nb < 128) nb = 128;
  if (growable) nb = 128;
  m->growable = growable;
  m->NB = nb;
  if (P < 1) { P = G.nranks; r = G.rank; } else { m->virtual_rank = true; }
  m->P = P;
  m->r = r;
  if (P > 1 && keep_k) { delete m; return gpk_set_err(-1, "keep-K models are single-GPU only"); }
  // Block-cyclic models take the SAME panel widths by remaining size as a one-GPU model (so that every element receives
  // the same rank-w updates in the same order: the P-rank factor is the one-GPU factor bit for bit under the default
  // schedule too, and the bulk updates run at K = 512 where the one-GPU ones do).  An ownership block is dist_group
  // panels of the middle width, and a whole number of the widest panels: no panel ever straddles two owners.
  const bool auto_w = G.nb_auto() && !growable;
  const int group = (P > 1) ? std::max(1, G.dist_group) : 1;
  m->OB = (P > 1) ? (auto_w ? round_up(group * nb, std::max(nb, G.nb_big)) : group * nb) : 0;
  m->Npad = (P > 1) ? round_up(N, m->OB) : round_up(N, GPK_T);
  if (P == 1) m->OB = m->Npad;
  m->nT = m->Npad / GPK_T;
  // panel layout (see gpk_model): inner widths by the remaining size (nb_auto) or uniform nb; outer panels group
  // the inner ones where the two-level schedule applies
  auto inner_width = [&](int rem) {
    int w = nb;
    if (auto_w) {
      if (rem > G.nb_big_until + G.nb_big) w = G.nb_big;
      else if (rem <= G.nb_small_from) w = G.nb_small;
    }
    if (P > 1) w = std::min(w, m->OB - (m->Npad - rem) % m->OB);  // (thresholds that are no multiples of the block)
    return w;
  };
  const bool two_level = (P == 1 && !growable && G.nb_auto() && G.lookahead && G.nb_outer >= 256 && m->Npad >= G.outer_from());
  size_t off = 0;
  for (int c0 = 0; c0 < m->Npad;) {
    const int rem = m->Npad - c0;
    int W = std::min(inner_width(rem), rem);
    if (two_level && rem > G.outer_until) W = std::min(G.nb_outer / 128 * 128, rem);
    const int o = (int)m->oc0.size();
    m->oc0.push_back(c0);
    m->ow.push_back(W);
    m->ooff.push_back(off);
    m->ofirst.push_back((int)m->pc0.size());
    for (int ci = c0; ci < c0 + W;) {
      const int w = std::min(inner_width(m->Npad - ci), c0 + W - ci);
      m->pc0.push_back(ci);
      m->pw.push_back(w);
      m->pld.push_back(W);
      m->pouter.push_back(o);
      m->poff.push_back(off + (size_t)(ci - c0) * W + (ci - c0));
      ci += w;
    }
    m->oend.push_back((int)m->pc0.size());
    off += (size_t)rem * W;
    c0 += W;
  }
  m->nO = (int)m->oc0.size();
  m->nP = (int)m->pc0.size();
  m->L_elems = off;
  // block-cyclic models: the chain-bound last columns as ONE tail block on one rank (gpk_model::ct), so that the tile pipeline can
  // run there (dist_tail_phase).  From the first 128-wide panel that starts an ownership block with at most tail_rows rows left.
  m->ct = 0;
  if (P > 1 && !growable && G.lookahead && G.tail_rows > 0 && G.dist_tail) {
    for (int p = 1; p < m->nP; ++p) {
      if (m->pw[p] != 128 || m->pc0[p] % m->OB || m->Npad - m->pc0[p] > G.tail_rows) continue;
      bool all128 = true;
      for (int q = p; q < m->nP; ++q) all128 = all128 && m->pw[q] == 128 && m->pld[q] == 128;
      if (all128 && m->nP - p >= 4) { m->ct = m->pc0[p]; m->p_tail = p; }
      break;
    }
  }
  // Who owns the tail, and does it take part in the cyclic map?  Work model (flops of the factorisation): the tail block costs
  // T^2 ct (every column in front of it updates its lower half) + T^3 / 3, the cyclic part the rest of Npad^3 / 3.  Owner inside
  // the map: it carries tail + cyclic / P.  Owner outside (rank P - 1, the others cyclic over P - 1): max(tail, cyclic / (P - 1)).
  m->Pc = P;
  m->t_own = m->ct > 0 ? (m->ct / m->OB) % P : 0;
  if (m->ct > 0 && P >= 3 && G.dist_tail_alone != 0) {
    const double T = m->Npad - m->ct, n3 = (double)m->Npad * m->Npad * m->Npad / 3.0;
    const double tail = T * T * m->ct + T * T * T / 3.0, cyc = n3 - tail;
    if (G.dist_tail_alone > 0 || std::max(tail, cyc / (P - 1)) < tail + cyc / P) {
      m->Pc = P - 1;
      m->t_own = P - 1;
    }
  }
  m->nloc = 0;
  for (int b = 0; b < (m->ct > 0 ? m->ct : m->Npad) / m->OB && r < m->Pc; ++b)
    if (b % m->Pc == r) m->nloc++;
  m->nloc_c = m->nloc;
  m->ldk = (P > 1) ? (long)m->nloc * m->OB + ((m->ct > 0 && r == m->t_own) ? m->Npad - m->ct : 0) : (long)m->Npad;
  if (m->ldk < m->OB) m->ldk = m->OB;
  m->keep_k = keep_k;
  const size_t np = m->Npad;
  int rc = 0;
  m->dX_elems = growable ? np * (size_t)d : (size_t)N * d;
  rc |= dev_alloc_t(&m->dX, m->dX_elems);
  if (!growable && d <= 8) {  // (a growing model's point set changes under the scaled copy)
    m->dXs_elems = se_scratch_doubles((int)np, 0, d, false);
    rc |= dev_alloc_t(&m->dXs, m->dXs_elems);
  }
  rc |= dev_alloc_t(&m->dy, np);
  rc |= dev_alloc_t(&m->dw, np);
  rc |= dev_alloc_t(&m->dK, np * (size_t)m->ldk);
  if (keep_k) rc |= dev_alloc_t(&m->dKcopy, np * np);
  rc |= dev_alloc_t(&m->dL, m->L_elems);
  rc |= dev_alloc_t(&m->dDinv, (size_t)m->nT * 128 * 128);
  rc |= dev_alloc_t(&m->dscal, 8);
  m->dws_bytes = (size_t)64 * std::max(nb, G.nb_big) * 8;
  rc |= dev_alloc(reinterpret_cast<void**>(&m->dws), m->dws_bytes);
  rc |= dev_alloc_t(&m->dinfo, 4);
  rc |= dev_alloc_t(&m->dflags, (size_t)3 * m->nT);
  if (!rc && hipMemsetAsync(m->dflags, 0, (size_t)3 * m->nT * sizeof(unsigned), G.s_main) != hipSuccess) rc = gpk_set_err(-2, "memset failed");
  if (rc) {
    model_destroy(m);
    return -2;
  }
  *out = m;
  return 0;
}

// Buffers and events go back in the REVERSE order of model_create / model_events, so that the next model of the same shape
// -- an optimiser evaluates one shape hundreds of times, GP_regression_demo.py:44 -- gets the very same pointers and event
// handles back from the pools field by field.
void model_destroy(gpk_model* m) {
  if (!m) return;
  const size_t np = m->Npad;
  dev_free(m->dflags, (size_t)3 * m->nT * sizeof(unsigned));
  dev_free(m->dinfo, 16);
  dev_free(m->dws, m->dws_bytes);
  dev_free(m->dscal, 64);
  dev_free(m->dDinv, (size_t)m->nT * 128 * 128 * 8);
  dev_free(m->dL, m->L_elems * 8);
  if (m->dKcopy) dev_free(m->dKcopy, np * np * 8);
  dev_free(m->dK, np * (size_t)m->ldk * 8);
  dev_free(m->dw, np * 8);
  dev_free(m->dy, np * 8);
  if (m->dZs) dev_free(m->dZs, m->dZs_elems * 8);
  if (m->dXs) dev_free(m->dXs, m->dXs_elems * 8);
  dev_free(m->dX, m->dX_elems * 8);
  if (m->hpin) G.hpin_pool[m->hpin_doubles].push_back(m->hpin);
  for (size_t i = m->ev.size(); i-- > 0;) G.ev_dep_free.push_back(m->ev[i]);
  delete m;
}

// GPr.py:90-97: hyp = exp(logHyp); M = hyp[:n-2]; sf2 = hyp[n-2]^2; sn2 = hyp[n-1]^2
int decode_hyp(const double* log_hyp, int d, double jitter, gpk_model* m) {
  m->se.d = d;
  for (int k = 0; k < d; ++k) {
    const double ell = std::exp(log_hyp[k]);
    if (!(ell > 0.0) || !std::isfinite(ell)) return gpk_set_err(-1, "length scale %d is not positive finite", k);
    m->se.ell[k] = ell;
  }
  const double sf = std::exp(log_hyp[d]), sn = std::exp(log_hyp[d + 1]);
  m->se.sf2 = sf * sf;
  m->sn2 = sn * sn;
  m->jitter = jitter;
  m->se.diag_add = m->sn2 + jitter;
  return 0;
}

// ------------------------------------------------------------------ blocked Cholesky
// Right-looking over outer panels of width NB (one MFMA SYRK per panel, K = NB keeps the
// C read-modify-write traffic at 8/NB bytes per flop), left-looking over the 128-wide
// sub-panels inside a panel.  L goes to its own panel-packed storage (panel p =
// rows c0.. x w, ld = w): compact operands for every later kernel and no in-place hazard
// in the panel solve.  Replaces np.linalg.cholesky (GPr.py:62, GPpref.py:128).
// rows_left: rows of the matrix from this tile on.  While much of the matrix remains the chip is full of trailing-update
// workgroups and the leaf takes the slot-sized form (it fits where ONE of them has just finished, and the GEMM workgroup it
// then shares the CU with pauses: gpk_internal.h); once few rows remain an empty CU is easy to find and the full-tile
// form, alone on its CU, is the faster one (option leaf_fat_rows).  Same bits either way.
int leaf_launch(const double* A, long lda, double* L, long ldl, double* Dinv, int* info, int col0, hipStream_t s, int rows_left) {
  return launch_potf2_trtri_128(A, lda, L, ldl, Dinv, info, col0, 0, s, rows_left <= G.leaf_fat_rows ? 1 : 0);
}

int factor_panel(gpk_model* m, int p, hipStream_t s) {
  const int c0 = m->pc0[p], w = m->pw[p], np_ = m->Npad;
  const long ldk = m->ldk, ld = m->pld[p];
  const long kc0 = m->local_col(c0);  // local column of global column c0 (the caller owns panel p)
  double* Lp = m->Lpanel(p);
  const int h = np_ - c0;
  for (int sb = 0; sb < w / 128; ++sb) {
    const int cs = c0 + sb * 128;   // global column of this sub-panel
    const int ro = cs - c0;         // row offset inside the panel
    double* Acol = m->dK + (long)cs * ldk + kc0 + sb * 128;  // A[cs, cs]
    if (sb > 0) {
      // A[cs:, cs:cs+128] -= Lp[ro:, 0:ro] * Lp[ro:ro+128, 0:ro]^T
      GemmNT g{};
      g.A = Lp + (long)ro * ld; g.lda = ld;
      g.B = Lp + (long)ro * ld; g.ldb = ld;
      g.C = Acol; g.ldc = ldk;
      g.m = h - ro; g.n = 128; g.k = ro; g.sub = 1;
      GPK_TRY(launch_gemm_nt(g, false, s));
    }
    double* Dt = m->dDinv + (long)(cs / 128) * 128 * 128;
    GPK_TRY(leaf_launch(Acol, ldk, Lp + (long)ro * ld + sb * 128, ld, Dt, m->dinfo, cs, s, h - ro));
    if (h - ro - 128 > 0) {
      // L[cs+128:, sub-panel] = A[cs+128:, cs:cs+128] * Dinv^T
      GemmNT g{};
      g.A = Acol + (long)128 * ldk; g.lda = ldk;
      g.B = Dt; g.ldb = 128;
      g.C = Lp + (long)(ro + 128) * ld + sb * 128; g.ldc = ld;
      g.m = h - ro - 128; g.n = 128; g.k = 128; g.sub = 0;
      GPK_TRY(launch_gemm_nt(g, false, s));
    }
  }
  return 0;
}

// A[r0g:, c0g : c0g+ncols] -= Lp[(r0g-c0p):, :] * Lp[(c0g-c0p) : +ncols, :]^T   (rows r0g.., global indices)
// lower = 1: square region on the diagonal (r0g == c0g, ncols == rows), only tiles on/below it.
int panel_update(gpk_model* m, int p, int r0g, int c0g, int ncols, int lower, hipStream_t s) {
  const int c0 = m->pc0[p], w = m->pw[p], np_ = m->Npad;
  const long ld = m->pld[p];
  const int rows = np_ - r0g;
  if (rows <= 0 || ncols <= 0) return 0;
  double* Lp = m->Lpanel(p);
  GemmNT g{};
  g.A = Lp + (long)(r0g - c0) * ld; g.lda = ld;
  g.B = Lp + (long)(c0g - c0) * ld; g.ldb = ld;
  g.C = m->dK + (long)r0g * m->ldk + m->local_col(c0g); g.ldc = m->ldk;
  g.m = rows; g.n = ncols; g.k = w; g.sub = 1; g.lower = lower;
  // Few 128x128 tiles = a latency-bound launch on the panel chain (look-ahead block column, tail of the
  // factorisation): 64x64 tiles put 4x as many CUs on it (84 tiles at K = 256: 70 us -> 64x64: ~20 us).
  const long nc = ncols / 128, nr = rows / 128;
  const long tiles = lower ? nc * (nc + 1) / 2 + (nr - nc) * nc : nr * nc;
  if (tiles < G.small_tile_below) return launch_gemm_nt(g, false, s);
  return launch_gemm_auto(g, s);
}

// The same for a whole OUTER panel o (K = ow[o], operands in the outer panel's packed storage).
int outer_update(gpk_model* m, int o, int r0g, int c0g, int ncols, int lower, hipStream_t s) {
  const int c0 = m->oc0[o], W = m->ow[o], np_ = m->Npad;
  const int rows = np_ - r0g;
  if (rows <= 0 || ncols <= 0) return 0;
  double* Lo = m->Lo(o);
  GemmNT g{};
  g.A = Lo + (long)(r0g - c0) * W; g.lda = W;
  g.B = Lo + (long)(c0g - c0) * W; g.ldb = W;
  g.C = m->dK + (long)r0g * m->ldk + m->local_col(c0g); g.ldc = m->ldk;
  g.m = rows; g.n = ncols; g.k = W; g.sub = 1; g.lower = lower;
  const long nc = ncols / 128, nr = rows / 128;
  const long tiles = lower ? nc * (nc + 1) / 2 + (nr - nc) * nc : nr * nc;
  if (tiles < G.small_tile_below) return launch_gemm_nt(g, false, s);
  return launch_gemm_auto(g, s);
}

// Distributed trailing update of panel p on THIS rank's columns whose global index lies in [gc_lo, gc_hi), rows from
// the end of panel p's diagonal block.  The kernel maps local tile columns to global columns (GemmNT.cyc_*) for the
// B-operand rows and the lower-triangle test.
int dist_update(gpk_model* m, int p, long gc_lo, long gc_hi, hipStream_t s) {
  const int c0 = m->pc0[p], w = m->pw[p], np_ = m->Npad;
  const int r0 = c0 + w;
  const int rows = np_ - r0;
  if (gc_lo < r0) gc_lo = r0;
  if (gc_hi > np_) gc_hi = np_;
  if (rows <= 0 || gc_hi <= gc_lo) return 0;
  double* Lp = m->Lpanel(p);
  if (m->ct > 0 && gc_hi > m->ct) {
    // the tail block (its owner only): a plain lower update from (cs, cs) -- rows above cs are above the diagonal of these columns
    if (m->r == m->t_own) {
      const long cs = std::max<long>(gc_lo, m->ct);
      const long nt = (np_ - cs) / 128, nc = (gc_hi - cs) / 128;
      GemmNT g{};
      g.A = Lp + (cs - c0) * m->pld[p]; g.lda = m->pld[p];
      g.B = g.A; g.ldb = g.lda;
      g.C = m->dK + cs * m->ldk + m->local_col(cs); g.ldc = m->ldk;
      g.m = (int)(np_ - cs); g.n = (int)(gc_hi - cs); g.k = w; g.sub = 1; g.lower = 1;
      if (nc * (nc + 1) / 2 + (nt - nc) * nc < G.small_tile_below) GPK_TRY(launch_gemm_nt(g, false, s));
      else GPK_TRY(launch_gemm_auto(g, s));
    }
    gc_hi = m->ct;
    if (gc_hi <= gc_lo) return 0;
  }
  const long lc_lo = m->local_from(gc_lo), lc_hi = m->local_from(gc_hi);
  if (lc_hi <= lc_lo) return 0;
  GemmNT g{};
  g.A = Lp + (long)w * m->pld[p]; g.lda = m->pld[p];
  g.B = g.A; g.ldb = g.lda;
  g.C = m->dK + (long)r0 * m->ldk + lc_lo; g.ldc = m->ldk;
  g.m = rows; g.n = (int)(lc_hi - lc_lo); g.k = w; g.sub = 1; g.lower = 1;
  g.cyc_P = m->Pc; g.cyc_r = m->r; g.cyc_NB = m->OB; g.cyc_lg0 = (int)(lc_lo / m->OB); g.cyc_c0 = (int)(lc_lo % m->OB);
  g.row_g0 = r0; g.brow_g0 = r0;
  // a rank's share of an update is 1/P of the tiles: the same latency rule as panel_update (every kernel of
  // k_gemm.hip rounds identically, so the factor does not depend on the choice)
  if ((long)(rows / 128) * (g.n / 128) < G.small_tile_below) return launch_gemm_nt(g, false, s);
  return launch_gemm_nt(g, true, s);
}
static double dist_update_flops(const gpk_model* m, int p, long gc_lo, long gc_hi) {
  double f = 0.0;
  const long r0 = m->pc0[p] + m->pw[p];
  for (long c = std::max(gc_lo, r0); c < std::min<long>(gc_hi, m->Npad); c += 128)
    if (m->owner_of_col(c) == m->r) f += (double)(m->Npad - c) * 128 * m->pw[p];
  return f;
}

int model_events(gpk_model* m, int n) {
  while ((int)m->ev.size() < n) {
    hipEvent_t e;
    if (!G.ev_dep_free.empty()) {
      e = G.ev_dep_free.back();
      G.ev_dep_free.pop_back();
    } else {
      GPK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    m->ev.push_back(e);
  }
  return 0;
}

// C[r0g:, r0g:] (lower) -= [Lp Lq][Lp Lq]^T for two CONSECUTIVE panels p, q = p+1 in ONE pass over C (K = w_p + w_q):
// half the C traffic and accumulator hand-overs per flop of two separate updates.  Falls back to two launches when
// the update is small enough for the 64x64-tile kernel (which has no second K segment).
static int pair_update(gpk_model* m, int p, int r0g, hipStream_t s) {
  const int np_ = m->Npad, rows = np_ - r0g;
  if (rows <= 0) return 0;
  const long nt = rows / 128;
  if (nt * (nt + 1) / 2 < G.small_tile_below) {
    GPK_TRY(panel_update(m, p, r0g, r0g, rows, 1, s));
    return panel_update(m, p + 1, r0g, r0g, rows, 1, s);
  }
  const int c0 = m->pc0[p], w = m->pw[p], c1 = m->pc0[p + 1], w1 = m->pw[p + 1];
  GemmNT g{};
  g.A = m->Lpanel(p) + (long)(r0g - c0) * m->pld[p]; g.lda = m->pld[p];
  g.B = g.A; g.ldb = g.lda;
  g.A2 = m->Lpanel(p + 1) + (long)(r0g - c1) * m->pld[p + 1]; g.lda2 = m->pld[p + 1];
  g.B2 = g.A2; g.ldb2 = g.lda2;
  g.k1 = w; g.k = w + w1;
  g.C = m->dK + (long)r0g * m->ldk + r0g; g.ldc = m->ldk;
  g.m = rows; g.n = rows; g.sub = 1; g.lower = 1;
  return launch_gemm_auto(g, s);
}

// Look-ahead factorisation with PAIRED trailing updates (single GPU).  Panels (p, p+1), p even:
//   s_panel:  LA(p -> block column p+1) . factor(p+1) . LA(p+1 -> block column p+2) . factor(p+2) ...
//   s_main :  near(p -> block column p+2) . farA(p, p+1 -> block column p+3) . farB(p, p+1 -> everything right of it)
// farB carries almost all the flops and runs with K = w_p + w_{p+1}; near/farA are the narrow pieces the panel chain
// needs early, so the chain runs up to two panels ahead of the bulk update and s_main never waits for a panel.
//   LA(p)   waits farA(p-2, p-1)   (block column p+1 complete w.r.t. all earlier panels)
//   LA(p+1) waits near(p)          (queued behind farB(p-2, p-1) on s_main, which covers block column p+2)
static int potrf_device_paired(gpk_model* m, const std::function<int(int)>* after_panel) {
  hipStream_t sm = G.s_main, sp = G.s_panel;
  const int nP = m->nP, np_ = m->Npad;
  GPK_TRY(model_events(m, 3 * nP + 6));
  g_potrf_lookahead_used = 1;
  hipEvent_t* ev_panel = m->ev.data();
  hipEvent_t* ev_near = m->ev.data() + nP;
  hipEvent_t* ev_farA = m->ev.data() + 2 * nP + 6;
  hipEvent_t ev_start = m->ev[2 * nP];
  auto c = [&](int q) { return q < nP ? m->pc0[q] : np_; };
  auto wd = [&](int q) { return q < nP ? m->pw[q] : 0; };
  GPK_HIP(hipEventRecord(ev_start, sm));
  GPK_HIP(hipStreamWaitEvent(sp, ev_start, 0));
  GPK_TRY(factor_panel(m, 0, sp));
  GPK_HIP(hipEventRecord(ev_panel[0], sp));
  if (G.stamps && G.dstamps) GPK_TRY(launch_stamp(G.dstamps + 0, sp));
  if (after_panel) GPK_TRY((*after_panel)(0));
  for (int p = 0; p < nP; p += 2) {
    const bool hasB = p + 1 < nP;
    // ---- panel chain: A = p
    if (hasB) {
      if (p >= 2) GPK_HIP(hipStreamWaitEvent(sp, ev_farA[p - 2], 0));
      GPK_TRY(panel_update(m, p, c(p + 1), c(p + 1), wd(p + 1), 0, sp));
      GPK_TRY(factor_panel(m, p + 1, sp));
      GPK_HIP(hipEventRecord(ev_panel[p + 1], sp));
      if (G.stamps && G.dstamps) GPK_TRY(launch_stamp(G.dstamps + p + 1, sp));
      if (after_panel) GPK_TRY((*after_panel)(p + 1));
    }
    // ---- update stream: near(p -> block column p+2)
    GPK_HIP(hipStreamWaitEvent(sm, ev_panel[p], 0));
    if (p + 2 < nP) GPK_TRY(panel_update(m, p, c(p + 2), c(p + 2), wd(p + 2), 0, sm));
    GPK_HIP(hipEventRecord(ev_near[p], sm));
    // ---- panel chain: B = p+1
    if (p + 2 < nP) {
      GPK_HIP(hipStreamWaitEvent(sp, ev_near[p], 0));
      GPK_TRY(panel_update(m, p + 1, c(p + 2), c(p + 2), wd(p + 2), 0, sp));
      GPK_TRY(factor_panel(m, p + 2, sp));
      GPK_HIP(hipEventRecord(ev_panel[p + 2], sp));
      if (G.stamps && G.dstamps) GPK_TRY(launch_stamp(G.dstamps + p + 2, sp));
      if (after_panel) GPK_TRY((*after_panel)(p + 2));
    }
    // ---- update stream: farA (block column p+3, from both panels), farB (the rest, one pass)
    if (hasB) GPK_HIP(hipStreamWaitEvent(sm, ev_panel[p + 1], 0));
    if (p + 3 < nP) {
      GPK_TRY(panel_update(m, p, c(p + 3), c(p + 3), wd(p + 3), 0, sm));
      GPK_TRY(panel_update(m, p + 1, c(p + 3), c(p + 3), wd(p + 3), 0, sm));
    }
    GPK_HIP(hipEventRecord(ev_farA[p], sm));
    if (p + 4 < nP) {
      const int r4 = c(p + 4);
      const long nt = (np_ - r4) / 128;
      const char* tname = (nt * (nt + 1) / 2 < G.small_tile_below) ? "trailing_small" : "trailing";
      timer_begin(tname, sm, (double)(np_ - r4) * (np_ - r4) * (wd(p) + wd(p + 1)));
      GPK_TRY(pair_update(m, p, r4, sm));
      timer_end(tname, sm);
    }
  }
  GPK_HIP(hipStreamWaitEvent(sm, ev_panel[nP - 1], 0));
  return 0;
}

// ------------------------------------------------------------------ tile pipeline for the chain-bound tail
// Where few columns remain the factorisation is a chain of dependent latency-bound kernels: per 128 columns a leaf
// (50 us) and two or three full-height skinny GEMMs, each of which waits for CU slots next to the running update
// (25-35 us under load instead of 8).  This schedule takes the full-height GEMMs OFF the chain.  With 128-wide panels
// and tile rows r = p+1, p+2, ... below diagonal tile p, step p is split over two streams (event names in brackets):
//   s_panel (critical):  leaf(p)                                              needs tile (p,p) final
//                        T(p): L(p+1,p) = A(p+1,p) Dinv(p)^T;  A(p+1,p+1) -= L(p+1,p) L(p+1,p)^T   [tt]
//                              two one-tile GEMMs; needs LAnear(p-1)
//   s_main, in front of the bulk:
//                        SOLVE(p): L(i,p) = A(i,p) Dinv(p)^T, i >= p+2       needs T(p) (only the leaf, really: one event
//                                                                             per step instead of two)            [panel]
//                        LAnear(p): A(i, p+1..p+2) -= L(i,p) L(p+1..p+2, p)^T, i >= p+2                            [near]
//                        LAfar(p):  A(i, p+3)      -= L(i,p) L(p+3,p)^T,       i >= p+3   (after REST(p-1): stream order)
//                        REST(p): lower square from tile (p+4,p+4), the bulk
// so the chain per step is leaf + two one-tile kernels, and a column is written, in this order, by REST (up to four
// steps earlier), LAfar, LAnear and T: every hazard is covered by stream order on s_main plus the two waits above.
// Each element receives the same rank-128 updates in the same order as in the panel-level schedule with 128-wide
// panels, so the factor is bit-identical to it (tests).  `ev_prev_rest`: what plays REST(p_first - 1).
// What was tried on top of this and dropped (a stream of its own for the full-height GEMMs, the two one-tile GEMMs in one
// launch, flag-chained steps without event packets, a resident leaf server): DESIGN.md section 6.
static int span_update(gpk_model* m, int o_first, int o_last, int r0g, int c0g, int ncols, int lower, hipStream_t s, int persist_grid);
int g_potrf_tail_stream = 0;  // 1 if the last tile pipeline ran its near GEMMs on the near stream
static int near_stream_ensure() {
  if (G.s_near) return 0;
  int lo = 0, hi = 0;
  GPK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  GPK_HIP(hipStreamCreateWithPriority(&G.s_near, hipStreamNonBlocking, hi));
  return 0;
}
static int potrf_tail(gpk_model* m, int p_first, hipEvent_t ev_prev_rest, const std::function<int(int)>* after_panel) {
  hipStream_t sm = G.s_main, sc = G.s_panel;
  const int nP = m->nP, np_ = m->Npad;
  const long ldk = m->ldk;
  hipEvent_t* ev_panel = m->ev.data();
  hipEvent_t* ev_leaf = m->ev.data() + 3 * nP + 6;
  hipEvent_t* ev_tt = m->ev.data() + 4 * nP + 6;
  hipEvent_t* ev_near = m->ev.data() + 5 * nP + 6;
  hipEvent_t* ev_far = m->ev.data() + 6 * nP + 6;
  if (ev_prev_rest) GPK_HIP(hipStreamWaitEvent(sm, ev_prev_rest, 0));
  // one launch per chain step (potf2_chain_kernel) where the full-tile leaf is used and cross-stream flags work
  const bool fused_ok = (G.chain_fused > 0 || (G.chain_fused < 0 && G.flags_ok)) && m->dflags;
  // The NEAR stream (option tail_stream): the full-height solve and the look-ahead update of tile columns p+1, p+2 -- what the
  // chain waits for next -- on a stream of their own instead of in front of the bulk on the update stream, where they queue
  // behind REST(p-1) (35 us at 4096 rows).  A fourth stream that does not share a hardware queue with one of the other three has
  // to be a HIGH-priority one on this runtime (profiles/r04_stream_probe_null.txt), and a fifth never fits: not in a process with
  // a communication stream.  Hazards across the two streams: near(p) writes tile columns p+1, p+2, whose earlier writers on the
  // update stream -- REST(p-3), REST(p-2), LAfar(p-2), LAfar(p-1) -- all precede the event `far(p-1)` recorded behind LAfar(p-1);
  // LAfar(p) and REST(p) read rows of panel p that the solve on the near stream writes: they wait for near(p).
  hipStream_t sn = sm;
  if (G.tail_stream && !G.s_comm && !fused_ok) {
    GPK_TRY(near_stream_ensure());
    sn = G.s_near;
    hipEvent_t ev_in = m->ev[2 * nP + 1];  // everything queued on the update stream so far (the hand-over's updates)
    GPK_HIP(hipEventRecord(ev_in, sm));
    GPK_HIP(hipStreamWaitEvent(sn, ev_in, 0));
  }
  g_potrf_tail_stream = sn != sm ? 1 : 0;
  const unsigned epoch = ++G.flag_epoch;
  g_potrf_chain_fused = 0;
  for (int p = p_first; p < nP; ++p) {
    const int c0 = m->pc0[p], h = np_ - c0;
    const long ld = m->pld[p];
    double* Lp = m->Lpanel(p);
    double* Acol = m->dK + (long)c0 * ldk + m->local_col(c0);  // (tile (p,p); the tail columns are contiguous in local storage too)
    double* Dt = m->dDinv + (long)(c0 / 128) * 128 * 128;
    // ---- critical stream.  One event per step: `tt` after the two one-tile GEMMs (or `leaf` for the last tile, which has
    // none): an event packet on the chain stream costs ~5 us of a ~85 us step, and the full-height solve is never what the
    // chain waits for next (near(p) is due one leaf later).
    const bool fused = fused_ok && h > 128 && h <= G.leaf_fat_rows;
    // ... and the update stream's full-height solve / look-ahead update of this step do not wait for that launch to END: they poll
    // the flags it raises after the leaf and after L(p+1,p) (chain_fused = 2; 1: they wait for the event behind the launch)
    const bool early = fused && (G.chain_fused == 2 || G.chain_fused < 0);
    if (fused) {
      // leaf(p), L(p+1,p) and the update of tile (p+1,p+1) in one launch; the kernel itself waits (after the leaf) for the flag the
      // update stream sets behind LAnear(p-1) -- the first step's row p+1 was brought up to date on this stream by the hand-over
      GPK_TRY(launch_potf2_chain(Acol, ldk, Lp, ld, Dt, m->dinfo, c0, p > p_first ? m->dflags + (p - 1) : nullptr, epoch, sc,
                                 early ? m->dflags + m->nT + p : nullptr, early ? m->dflags + 2 * m->nT + p : nullptr));
      GPK_HIP(hipEventRecord(ev_tt[p], sc));
      ++g_potrf_chain_fused;
    } else {
    GPK_TRY(leaf_launch(Acol, ldk, Lp, ld, Dt, m->dinfo, c0, sc, h));
    if (h <= 128) GPK_HIP(hipEventRecord(ev_leaf[p], sc));
    }
    if (h > 128 && !fused) {
      if (p > p_first) GPK_HIP(hipStreamWaitEvent(sc, ev_near[p - 1], 0));
      GemmNT t1{};  // L(p+1,p) = A(p+1,p) Dinv^T
      t1.A = Acol + (long)128 * ldk; t1.lda = ldk;
      t1.B = Dt; t1.ldb = 128;
      t1.C = Lp + (long)128 * ld; t1.ldc = ld;
      t1.m = 128; t1.n = 128; t1.k = 128; t1.sub = 0;
      GemmNT t2{};  // A(p+1,p+1) -= L(p+1,p) L(p+1,p)^T
      t2.A = Lp + (long)128 * ld; t2.lda = ld;
      t2.B = t2.A; t2.ldb = ld;
      t2.C = Acol + (long)128 * ldk + 128; t2.ldc = ldk;
      t2.m = 128; t2.n = 128; t2.k = 128; t2.sub = 1; t2.lower = 1;
      // (automatic: where the whole factorisation is chain-bound, Npad < 12288 -- N = 4096 -6 %, 2048 -5 %, 8192 -2 %; at N = 16384
      // the pipeline's steps are set by the substitution's catch-up GEMMs, the 64 priority waves of this kernel only delay those: +0.5 %)
      if (G.tail_tile_kernel > 0 || (G.tail_tile_kernel < 0 && np_ < 12288)) {
        GPK_TRY(launch_gemm_tile128(t1, sc));
        GPK_TRY(launch_gemm_tile128(t2, sc));
      } else {
        GPK_TRY(launch_gemm_nt(t1, false, sc));
        GPK_TRY(launch_gemm_nt(t2, false, sc));
      }
      GPK_HIP(hipEventRecord(ev_tt[p], sc));
    }
    // ---- full-height panel GEMMs, in front of the bulk on the update stream
    if (!early) GPK_HIP(hipStreamWaitEvent(sn, h <= 128 ? ev_leaf[p] : ev_tt[p], 0));
    if (sn != sm && p > p_first) GPK_HIP(hipStreamWaitEvent(sn, ev_far[p - 1], 0));
    if (h > 256) {
      GemmNT g{};  // SOLVE(p): rows from tile p+2
      g.A = Acol + (long)256 * ldk; g.lda = ldk;
      g.B = Dt; g.ldb = 128;
      g.C = Lp + (long)256 * ld; g.ldc = ld;
      g.m = h - 256; g.n = 128; g.k = 128; g.sub = 0;
      if (early) { g.wait_flag = m->dflags + m->nT + p; g.wait_val = epoch; g.wait_info = m->dinfo; }
      GPK_TRY(launch_gemm_nt(g, false, sn));
    }
    if (early && h <= 256) GPK_HIP(hipStreamWaitEvent(sm, ev_tt[p], 0));  // (no polling launch in this step: the event after all)
    // every row of panel p is in place (early: not before the look-ahead update below, which is what has waited for L(p+1,p))
    auto panel_done = [&]() -> int {
      GPK_HIP(hipEventRecord(ev_panel[p], sn));
      if (G.stamps && G.dstamps) GPK_TRY(launch_stamp(G.dstamps + p, sn));
      if (after_panel) GPK_TRY((*after_panel)(p));
      return 0;
    };
    const bool late_panel_event = early && h > 256;
    if (!late_panel_event) GPK_TRY(panel_done());
    if (h > 256) {
      GemmNT g{};  // LAnear(p): tile columns p+1, p+2, rows from tile p+2
      g.A = Lp + (long)256 * ld; g.lda = ld;
      g.B = Lp + (long)128 * ld; g.ldb = ld;
      g.C = Acol + (long)256 * ldk + 128; g.ldc = ldk;
      g.m = h - 256; g.n = std::min(256, h - 128); g.k = 128; g.sub = 1;
      if (early) { g.wait_flag = m->dflags + 2 * m->nT + p; g.wait_val = epoch; g.wait_info = m->dinfo; }
      GPK_TRY(launch_gemm_nt(g, false, sn));
    }
    if (late_panel_event) GPK_TRY(panel_done());
    GPK_HIP(hipEventRecord(ev_near[p], sn));
    if (sn != sm) GPK_HIP(hipStreamWaitEvent(sm, ev_near[p], 0));  // LAfar(p), REST(p) read the rows the solve has written (and the update stream joins the near stream step by step)
    if (fused_ok && h > 256) GPK_TRY(launch_flag_set(m->dflags + p, epoch, sm));  // what the fused step p+1 polls
    if (h > 384) {
      GemmNT g{};  // LAfar(p): tile column p+3, rows from tile p+3
      g.A = Lp + (long)384 * ld; g.lda = ld;
      g.B = g.A; g.ldb = ld;
      g.C = Acol + (long)384 * ldk + 384; g.ldc = ldk;
      g.m = h - 384; g.n = 128; g.k = 128; g.sub = 1; g.lower = 1;
      GPK_TRY(launch_gemm_nt(g, false, sm));
    }
    if (sn != sm) GPK_HIP(hipEventRecord(ev_far[p], sm));
    // ---- bulk
    if (h > 512) {
      const int r1 = c0 + 512;
      const long nt = (np_ - r1) / 128;
      const char* tname = (nt * (nt + 1) / 2 < G.small_tile_below) ? "trailing_small" : "trailing";
      timer_begin(tname, sm, (double)(np_ - r1) * (np_ - r1) * 128);
      GPK_TRY(panel_update(m, p, r1, r1, np_ - r1, 1, sm));
      timer_end(tname, sm);
    }
  }
  // join: everything after potrf runs on s_main (which has waited for every step's chain event)
  return 0;
}

// ------------------------------------------------------------------ lazy corner (see Runtime::corner_from)
int g_potrf_corner[3] = {0, 0, 0};
int corner_stream_ensure() {
#ifndef GPK_HOSTSIM
  if (G.s_corner && G.s_corner_prio == G.corner_prio) return 0;
  if (G.s_corner) {
    GPK_HIP(hipStreamSynchronize(G.s_corner));
    GPK_HIP(hipStreamDestroy(G.s_corner));
    G.s_corner = nullptr;
  }
  int lo = 0, hi = 0;
  GPK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  const int pr = G.corner_prio >= 2 ? hi : (G.corner_prio == 1 ? (lo + hi) / 2 : lo);
  GPK_HIP(hipStreamCreateWithPriority(&G.s_corner, hipStreamNonBlocking, pr));
  G.s_corner_prio = G.corner_prio;
#else
  if (!G.s_corner) GPK_HIP(hipStreamCreateWithPriority(&G.s_corner, hipStreamNonBlocking, 0));
#endif
  return 0;
}
static int tail_first_panel(const gpk_model* m);
// The boundaries (options corner_from, corner_from2, corner_from3; increasing): panel starts, the first with at least two panels in
// front of it, none beyond the start of the tile-pipeline tail, at least two tile columns behind each.  One-level schedules on
// one GPU only (with two-level blocking the outer updates already run at K = nb_outer).  out3: the boundaries found (0-filled).
int corner_boundaries(const gpk_model* m, int* out3) {
  out3[0] = out3[1] = out3[2] = 0;
  const int wants[3] = {G.corner_from < 0 ? 0 : G.corner_from, G.corner_from2, G.corner_from3};  // (automatic: off until DESIGN.md section 6 says otherwise)
  if (wants[0] <= 0 || m->P != 1 || m->growable || !G.lookahead || m->nO != m->nP || m->nP < 4) return 0;
  const int o_tail = tail_first_panel(m);
  const int cmax = (o_tail < m->nO) ? m->oc0[o_tail] : m->Npad - 256;
  int n = 0, o = 2;
  for (int i = 0; i < 3 && wants[i] > 0; ++i) {
    while (o < m->nO && m->oc0[o] < wants[i]) ++o;
    if (o >= m->nO || m->oc0[o] > cmax || m->Npad - m->oc0[o] < 256) break;
    out3[n++] = m->oc0[o];
    ++o;  // the next boundary is a later panel start
  }
  return n;
}
int corner_boundary(const gpk_model* m) {
  int b[3];
  corner_boundaries(m, b);
  return b[0];
}
// Region i of the lazy schedule: columns [b, bend) x rows >= b (a lower trapezoid).  It receives the updates of the panels in front
// of b from catch-up launches on the corner stream (chunks of panels, in order; none before panel hold_panel has been factored),
// the last few of them -- panels [final_first, q) -- together with the update of the boundary panel q - 1.
struct CornerRegion {
  int b = 0, bend = 0, q = 0;
  int final_first = 0;
  std::vector<int> chunk_end;   // chunk j = panels [chunk_end[j-1], chunk_end[j]), chunk_end.back() == final_first
  int hold_panel = 0;
  size_t next_chunk = 0;
  hipEvent_t ev_last = nullptr;  // the last catch-up launch queued so far
  int ev_base = 0;               // this region's first event in m->ev
};
static std::vector<CornerRegion> corner_plan(const gpk_model* m) {
  std::vector<CornerRegion> R;
  int bs[3];
  const int n = corner_boundaries(m, bs);
  const int chunk = std::max(G.corner_chunk, 128);
  for (int i = 0; i < n; ++i) {
    CornerRegion r;
    r.b = bs[i];
    r.bend = (i + 1 < n) ? bs[i + 1] : m->Npad;
    while (m->pc0[r.q] < r.b) ++r.q;
    int f = r.q - 1;
    while (f > 0 && r.b - m->pc0[f - 1] <= std::max(G.corner_last, m->pw[r.q - 1]) && r.q - (f - 1) <= GPK_MAXSEG) --f;
    r.final_first = f;
    int k = 0, cnt = 0;
    for (int p = 0; p < f; ++p) {
      k += m->pw[p];
      ++cnt;
      if (k >= chunk || cnt == GPK_MAXSEG || p == f - 1) {
        r.chunk_end.push_back(p + 1);
        k = cnt = 0;
      }
    }
    // the first region's catch-up is held back until the chain reaches column corner_hold; a later region's until the chain
    // reaches the boundary in front of it (its deferred work fills the chain-bound stretch between the two boundaries)
    const int hold_col = (i == 0) ? G.corner_hold : bs[i - 1];
    while (r.hold_panel + 1 < f && m->pc0[r.hold_panel] < hold_col) ++r.hold_panel;
    if (f > 0) r.hold_panel = std::min(r.hold_panel, f - 1);
    r.ev_base = (7 + i) * m->nP + 8;
    R.push_back(r);
  }
  return R;
}
// C[r0g:, c0g : c0g + ncols] -= sum over outer panels o in [o_first, o_last] of Lo[(r0g - oc0):, :] Lo[(c0g - oc0) : +ncols, :]^T, in
// panel order, as ONE launch of the 8-wave kernel over a K-segment table (one panel: the plain update, any tile size)
static int span_update(gpk_model* m, int o_first, int o_last, int r0g, int c0g, int ncols, int lower, hipStream_t s, int persist_grid) {
  if (o_first == o_last && persist_grid <= 0) return outer_update(m, o_last, r0g, c0g, ncols, lower, s);
  const int rows = m->Npad - r0g;
  if (rows <= 0 || ncols <= 0) return 0;
  if (o_last - o_first + 1 > GPK_MAXSEG) return gpk_set_err(-1, "internal: %d panels in one update", o_last - o_first + 1);
  GemmSegs sg{};
  for (int o = o_first; o <= o_last; ++o) {
    const int W = m->ow[o];
    GemmSeg& q = sg.s[sg.n++];
    q.A = m->Lo(o) + (long)(r0g - m->oc0[o]) * W;
    q.B = m->Lo(o) + (long)(c0g - m->oc0[o]) * W;
    q.lda = q.ldb = W;
    q.k = W;
  }
  GemmNT g{};
  g.C = m->dK + (long)r0g * m->ldk + m->local_col(c0g); g.ldc = m->ldk;
  g.m = rows; g.n = ncols; g.sub = 1; g.lower = lower;
  unsigned* ctr = nullptr;
  if (persist_grid > 0 && G.dctr) ctr = G.dctr + (G.dctr_next++ & 63);
  return launch_gemm_nt_segs(g, sg, s, ctr ? persist_grid : 0, ctr);
}

// first outer panel of the tile-pipeline tail (nO: no tail): from there on every panel is a single 128-wide one
static int tail_first_panel(const gpk_model* m) {
  if (!G.tail_rows || m->P != 1) return m->nO;
  int o = m->nO;
  while (o > 0 && m->ow[o - 1] == 128 && m->oend[o - 1] - m->ofirst[o - 1] == 1 && m->Npad - m->oc0[o - 1] <= G.tail_rows) --o;
  if (m->nO - o < 4) return m->nO;  // not worth a stream hand-over
  return o;
}

// Right-looking with one panel of look-ahead: while the big SYRK of panel p runs on s_main, the
// high-priority s_panel stream updates block column p+1 and factors panel p+1, so the latency-bound
// leaf/panel kernels leave the critical path.
int potrf_device(gpk_model* m, const std::function<int(int)>* after_panel) {
  hipStream_t sm = G.s_main, sp = G.s_panel;
  if (m->P > 1) {
    if (m->virtual_rank) return gpk_set_err(-1, "internal: virtual ranks are factored as a group (potrf_group)");
    gpk_model* one = m;
    return potrf_group(&one, 1);
  }
  GPK_HIP(hipMemsetAsync(m->dinfo, 0, 16, sm));
  const int nP = m->nP;
  g_last_pc0 = m->pc0;
  g_last_pw = m->pw;
  g_potrf_lookahead_used = 0;
  if (!G.lookahead || nP < 3) {
    for (int p = 0; p < nP; ++p) {
      timer_begin("panel", sm);
      GPK_TRY(factor_panel(m, p, sm));
      timer_end("panel", sm);
      const int r0 = m->pc0[p] + m->pw[p];
      timer_begin("trailing", sm, (double)(m->Npad - r0) * (m->Npad - r0) * m->pw[p]);
      GPK_TRY(panel_update(m, p, r0, r0, m->Npad - r0, 1, sm));
      timer_end("trailing", sm);
    }
    return 0;
  }
  // paired updates let the panel chain run two panels ahead of the bulk update: +3-9 % where the chain is the bound
  // (N = 4096: 309 -> 318 posteriors/s, N = 8192: +2 %).  From N = 16384 on they lose: the K-doubled far update does
  // run faster (50 vs 44.5 TFLOP/s) but the three narrow updates per pair it needs (16 % of the flops) cost more
  // than that (N = 16384: 38.1-38.5 vs 36.9-37.05 ms, N = 32768: 241.8 vs 231.9).  1: only up to 8192, 2: always.
  g_potrf_tail_from = -1;
  g_potrf_corner[0] = g_potrf_corner[1] = g_potrf_corner[2] = 0;
  if (nP >= 6 && m->nO == nP && tail_first_panel(m) == m->nO && (G.pair_update == 2 || (G.pair_update == 1 && m->Npad <= 8192)))
    return potrf_device_paired(m, after_panel);
  GPK_TRY(model_events(m, 10 * nP + 8));
  g_potrf_lookahead_used = 1;
  hipEvent_t* ev_panel = m->ev.data();        // inner panel p factored (recorded on s_panel)
  hipEvent_t* ev_trail = m->ev.data() + nP;   // rest-update of OUTER step o done (recorded on s_main)
  hipEvent_t ev_start = m->ev[2 * nP];
  const int nO = m->nO, np_ = m->Npad;
  hipEvent_t* ev_blk = m->ev.data() + 2 * nP + 6;  // outer step o: block column o+1 fully updated (s_main)
  // Factor the block column of outer panel o on stream s: right-looking over its inner panels, each followed by
  // its update of the remaining columns OF THIS BLOCK COLUMN (K = inner width).  One level: just factor_panel.
  // `first_update_after`: what the first of those updates must wait for (the rest of the block column receives the
  // previous outer panel's update on s_main while the first inner panel is being factored here).
  auto factor_outer = [&](int o, hipStream_t s, hipEvent_t first_update_after) -> int {
    const int cend = m->oc0[o] + m->ow[o];
    for (int p = m->ofirst[o]; p < m->oend[o]; ++p) {
      GPK_TRY(factor_panel(m, p, s));
      GPK_HIP(hipEventRecord(ev_panel[p], s));
      if (G.stamps && G.dstamps) GPK_TRY(launch_stamp(G.dstamps + p, s));
      if (after_panel) GPK_TRY((*after_panel)(p));
      const int r0 = m->pc0[p] + m->pw[p];
      if (cend - r0 > 0) {
        if (p == m->ofirst[o] && first_update_after) GPK_HIP(hipStreamWaitEvent(s, first_update_after, 0));
        GPK_TRY(panel_update(m, p, r0, r0, cend - r0, 1, s));
      }
    }
    return 0;
  };
  // s_panel must not start before the K build queued on s_main has finished -- or, where the caller built block column 0 first
  // (gpk_model::ev_kfirst), before that: panel 0 only reads its own block column; everything after it waits for the whole build
  const int o_tail = tail_first_panel(m);  // outer panels from here on go through the tile pipeline
  const bool early0 = m->ev_kfirst && o_tail > 0;
  GPK_HIP(hipEventRecord(ev_start, sm));
  GPK_HIP(hipStreamWaitEvent(sp, early0 ? m->ev_kfirst : ev_start, 0));
  g_potrf_tail_from = (o_tail < nO) ? m->ofirst[o_tail] : -1;
  if (o_tail == 0) return potrf_tail(m, 0, nullptr, after_panel);
  // Lazy corner (Runtime::corner_from): a panel in front of a boundary b updates columns < b only; the region behind the boundary
  // gets the updates of all panels in front of it from catch-up launches on the corner stream (chunks of panels, in order), the
  // last few of them together with the update of the boundary panel, the first one that needs the region up to date.  Up to
  // three nested regions.  Every element still receives the same rank-w updates in the same order inside fp64 MFMA chains that
  // start from the stored value: the same bits.
  std::vector<CornerRegion> regs = corner_plan(m);
  for (int i = 0; i < 3; ++i) g_potrf_corner[i] = (i < (int)regs.size()) ? regs[i].b : 0;
  if (!regs.empty()) GPK_TRY(corner_stream_ensure());
  GPK_TRY(factor_outer(0, sp, nullptr));
  if (early0) GPK_HIP(hipStreamWaitEvent(sp, ev_start, 0));
  for (int o = 0; o < nO; ++o) {
    const int r0 = m->oc0[o] + m->ow[o];             // first row/col of the trailing matrix
    CornerRegion* bd = nullptr;                       // the region this panel's update brings up to date (it ends at that boundary)
    int ccap = np_;                                   // columns this panel's update may touch now: [.., ccap)
    for (auto& rg : regs) {
      if (o == rg.q - 1) { bd = &rg; ccap = rg.bend; break; }
      if (r0 < rg.b) { ccap = rg.b; break; }
    }
    const int pf = bd ? bd->final_first : o;         // the panels this step applies: [pf, o]
    hipEvent_t ev_bd = bd ? bd->ev_last : nullptr;    // (assigned below, after this step's catch-up launches are queued)
    int kspan = 0;
    for (int q = pf; q <= o; ++q) kspan += m->ow[q];
    // catch-up launches that are due: a chunk's panels are all factored (the last of them in the previous step) and the chain
    // has reached the region's hold column
    for (auto& rg : regs)
      while (rg.next_chunk < rg.chunk_end.size() && std::max(rg.chunk_end[rg.next_chunk] - 1, rg.hold_panel) <= o) {
        const int first = rg.next_chunk ? rg.chunk_end[rg.next_chunk - 1] : 0, last = rg.chunk_end[rg.next_chunk] - 1;
        hipStream_t sc = G.s_corner;
        GPK_HIP(hipStreamWaitEvent(sc, ev_panel[m->oend[std::max(last, rg.hold_panel)] - 1], 0));
        int kc = 0;
        for (int q = first; q <= last; ++q) kc += m->ow[q];
        const double mm = (double)(np_ - rg.b), below = (double)(np_ - rg.bend);
        timer_begin("trailing", sc, (mm * mm - below * below) * kc);
        GPK_TRY(span_update(m, first, last, rg.b, rg.b, rg.bend - rg.b, 1, sc, G.corner_grid));
        timer_end("trailing", sc);
        hipEvent_t e = m->ev[rg.ev_base + rg.next_chunk];
        GPK_HIP(hipEventRecord(e, sc));
        if (G.stamps && G.dstamps && &rg == &regs[0] && rg.next_chunk < 63) GPK_TRY(launch_stamp(G.dstamps + 448 + rg.next_chunk, sc));
        rg.ev_last = e;
        ++rg.next_chunk;
      }
    ev_bd = bd ? bd->ev_last : nullptr;
    if (o_tail < nO && o + 1 == o_tail) {
      // hand-over to the tile pipeline: its first leaf needs tile column o+1, its first T / LAnear the two columns
      // after it, so the chain stream applies this panel to three tile columns, the update stream to the rest
      // (which then stands for REST(first tail panel - 1))
      const int w_la = std::min(384, np_ - r0);
      hipStream_t srest = sm;
      if (o > 0) GPK_HIP(hipStreamWaitEvent(sp, ev_trail[o - 1], 0));
      if (ev_bd) GPK_HIP(hipStreamWaitEvent(sp, ev_bd, 0));
      GPK_TRY(span_update(m, pf, o, r0, r0, w_la, 1, sp, 0));
      GPK_HIP(hipStreamWaitEvent(srest, ev_panel[m->oend[o] - 1], 0));
      if (ev_bd) GPK_HIP(hipStreamWaitEvent(srest, ev_bd, 0));
      const int r1 = r0 + w_la;
      if (np_ - r1 > 0) {
        const long nt = (np_ - r1) / 128;
        const char* tname = (nt * (nt + 1) / 2 < G.small_tile_below) ? "trailing_small" : "trailing";
        timer_begin(tname, srest, (double)(np_ - r1) * (np_ - r1) * kspan);
        GPK_TRY(span_update(m, pf, o, r1, r1, np_ - r1, 1, srest, 0));
        timer_end(tname, srest);
      }
      GPK_HIP(hipEventRecord(ev_trail[o], srest));
      return potrf_tail(m, m->ofirst[o_tail], nullptr, after_panel);  // (the hand-over's rest update sits in front of the tail's launches on the update stream: stream order)
    }
    const int wn = (o + 1 < nO) ? m->ow[o + 1] : 0;  // width of the next block column
    const int w1 = (o + 1 < nO) ? m->pw[m->ofirst[o + 1]] : 0;  // ... of its first inner panel: the look-ahead column
    if (o + 1 < nO) {
      // (a) s_panel: the look-ahead column gets outer panel o's update (lower trapezoid) ...
      if (o > 0) GPK_HIP(hipStreamWaitEvent(sp, ev_trail[o - 1], 0));  // earlier updates of that column
      if (ev_bd) GPK_HIP(hipStreamWaitEvent(sp, ev_bd, 0));  // ... those of the region's catch-up launches
      GPK_TRY(span_update(m, pf, o, r0, r0, w1, 1, sp, 0));
    }
    GPK_HIP(hipStreamWaitEvent(sm, ev_panel[m->oend[o] - 1], 0));
    if (ev_bd) GPK_HIP(hipStreamWaitEvent(sm, ev_bd, 0));
    const long ntr = (np_ - r0 - w1) / 128;
    const char* tname = (ntr * (ntr + 1) / 2 < G.small_tile_below) ? "trailing_small" : "trailing";
    if (wn - w1 > 0) {
      // (b1) s_main: the other columns of block column o+1 first, so that the inner updates of its factorisation
      // can follow the first inner panel without waiting for the whole trailing update  (two-level schedules: no lazy corner)
      timer_begin(tname, sm, (double)(wn - w1) * (2.0 * (np_ - r0 - w1) - (wn - w1)) * m->ow[o]);
      GPK_TRY(outer_update(m, o, r0 + w1, r0 + w1, wn - w1, 1, sm));
      timer_end(tname, sm);
      GPK_HIP(hipEventRecord(ev_blk[o], sm));
    }
    //     ... and block column o+1 is factored on s_panel.
    if (o + 1 < nO) GPK_TRY(factor_outer(o + 1, sp, (wn - w1 > 0) ? ev_blk[o] : nullptr));
    // (b2) s_main: the rest of the trailing matrix (lower triangle starting after block column o+1; lazy: its columns below b)
    const int r1 = r0 + wn;
    if (ccap - r1 > 0) {
      // "trailing" = the launches of the 8-wave kernel (the roofline object of bench.py); the latency-bound
      // tail that outer_update routes to the 64x64-tile kernel is timed under its own name
      const double mm = (double)(np_ - r1), below = (double)(np_ - ccap);
      timer_begin(tname, sm, (mm * mm - below * below) * kspan);
      GPK_TRY(span_update(m, pf, o, r1, r1, ccap - r1, 1, sm, 0));
      timer_end(tname, sm);
    }
    GPK_HIP(hipEventRecord(ev_trail[o], sm));
  }
  // join: everything after potrf runs on s_main and needs the last panel
  GPK_HIP(hipStreamWaitEvent(sm, ev_panel[nP - 1], 0));
  return 0;
}

// One process per GPU, panels owned block-cyclically.  Step p: the owner factors panel p from its local
// block column, the factored panel (and its inverted diagonal tiles) is broadcast with RCCL over xGMI,
// every rank updates the block columns it owns.  Look-ahead: the owner of panel p+1 updates THAT block
// column first, factors it and starts its broadcast on the panel stream while all ranks run the rest
// of update p on the main stream.
//
// The schedule is written as per-rank phases (dist_*_phase) driven by potrf_group over the rank models that live
// in this process: one for a real rank, all P for the single-process harness of the tests, whose "broadcast" is a
// device copy from the owner's model.  Both run the very same phases in the same order.
static inline int panel_owner(const gpk_model* m, int p) { return m->owner_of_col(m->pc0[p]); }
// panels are broadcast on a stream of their own, so that the owner's chain goes on while its panel travels:
// the communication stream of a multi-rank process, the substitution stream in the single-process harness
static inline hipStream_t comm_stream() { return G.s_comm ? G.s_comm : (G.s_pred ? G.s_pred : G.s_panel); }

static int bcast_panel(gpk_model* m, int p, hipStream_t s) {
  const int c0 = m->pc0[p], w = m->pw[p], owner = panel_owner(m, p);
  const size_t cnt = (size_t)(m->Npad - c0) * w;
  const size_t dcnt = (size_t)(w / 128) * 128 * 128;
  double* dinv = m->dDinv + (size_t)(c0 / 128) * 128 * 128;
  if (m->virtual_rank) {
    if (!m->vgroup) return gpk_set_err(-1, "internal: virtual rank %d without a group", m->r);
    if (owner == m->r) return 0;
    const gpk_model* o = m->vgroup[owner];  // the caller has made `s` wait for the owner's factorisation of panel p
    GPK_HIP(hipMemcpyAsync(m->Lpanel(p), o->Lpanel(p), cnt * 8, hipMemcpyDeviceToDevice, s));
    GPK_HIP(hipMemcpyAsync(dinv, o->dDinv + (size_t)(c0 / 128) * 128 * 128, dcnt * 8, hipMemcpyDeviceToDevice, s));
    return 0;
  }
  GPK_TRY(comm_bcast(m->Lpanel(p), cnt, owner, s));
  GPK_TRY(comm_bcast(dinv, dcnt, owner, s));
  return 0;
}
static inline bool dist_lookahead(const gpk_model* m) { return G.lookahead && m->nP >= 3; }

// ---- look-ahead schedule, one rank.  Events of rank model m: ev[p] = panel p AVAILABLE on this rank (the owner: its
//      factorisation on s_panel; the others: the broadcast on the communication stream), ev[nP + p] = update p of the
//      columns right of the look-ahead column done (s_main), ev[2 nP] = start, ev[2 nP + 6 + p] = broadcast of panel
//      p finished on this rank (communication stream; only ordering between broadcasts and the final join need it).
static int dist_publish_panel(gpk_model* m, int p) {
  // every rank takes part in the broadcast of panel p, in panel order, on the communication stream
  hipStream_t sp = G.s_panel, sc = comm_stream();
  hipEvent_t* ev_avail = m->ev.data();
  hipEvent_t* ev_bcast = m->ev.data() + 2 * m->nP + 6;
  const int owner = panel_owner(m, p);
  if (owner == m->r) {
    GPK_HIP(hipEventRecord(ev_avail[p], sp));               // factored here: available at once
    GPK_HIP(hipStreamWaitEvent(sc, ev_avail[p], 0));
  } else if (m->virtual_rank) {
    GPK_HIP(hipStreamWaitEvent(sc, m->vgroup[owner]->ev[p], 0));  // the copy reads the owner's buffers
  }
  GPK_TRY(bcast_panel(m, p, sc));
  GPK_HIP(hipEventRecord(ev_bcast[p], sc));
  if (owner != m->r) GPK_HIP(hipEventRecord(ev_avail[p], sc));  // received
  return 0;
}
static int dist_la_begin(gpk_model* m) {
  hipStream_t sm = G.s_main, sp = G.s_panel, sc = comm_stream();
  const int nP = m->nP;
  GPK_TRY(model_events(m, 10 * nP + 8));  // (the tail's tile pipeline uses the one-GPU event slots)
  GPK_HIP(hipMemsetAsync(m->dinfo, 0, 16, sm));
  GPK_HIP(hipEventRecord(m->ev[2 * nP], sm));  // nothing may start before the K build queued on s_main
  GPK_HIP(hipStreamWaitEvent(sp, m->ev[2 * nP], 0));
  GPK_HIP(hipStreamWaitEvent(sc, m->ev[2 * nP], 0));
  if (panel_owner(m, 0) == m->r) GPK_TRY(factor_panel(m, 0, sp));
  return dist_publish_panel(m, 0);
}
// panel phase of step p: panel p+1 becomes available on this rank
static int dist_la_panel_phase(gpk_model* m, int p) {
  hipStream_t sp = G.s_panel;
  const int nP = m->nP;
  if (p + 1 >= nP) return 0;
  hipEvent_t* ev_avail = m->ev.data();
  hipEvent_t* ev_trail = m->ev.data() + nP;
  if (panel_owner(m, p + 1) == m->r) {
    const long c1 = m->pc0[p + 1];
    if (p > 0) GPK_HIP(hipStreamWaitEvent(sp, ev_trail[p - 1], 0));  // earlier updates of that column ran on s_main
    GPK_HIP(hipStreamWaitEvent(sp, ev_avail[p], 0));                   // panel p: own factorisation or received
    GPK_TRY(dist_update(m, p, c1, c1 + m->pw[p + 1], sp));
    GPK_TRY(factor_panel(m, p + 1, sp));
  }
  return dist_publish_panel(m, p + 1);
}
// update phase of step p (s_main): every owned column right of the look-ahead column gets update p
static int dist_la_update_phase(gpk_model* m, int p) {
  hipStream_t sm = G.s_main;
  const int nP = m->nP;
  hipEvent_t* ev_avail = m->ev.data();
  hipEvent_t* ev_trail = m->ev.data() + nP;
  const long from = (p + 1 < nP) ? m->pc0[p + 1] + m->pw[p + 1] : m->Npad;  // the look-ahead column is the chain's
  GPK_HIP(hipStreamWaitEvent(sm, ev_avail[p], 0));
  timer_begin("trailing", sm, dist_update_flops(m, p, from, m->Npad));
  GPK_TRY(dist_update(m, p, from, m->Npad, sm));
  timer_end("trailing", sm);
  GPK_HIP(hipEventRecord(ev_trail[p], sm));
  return 0;
}
// panels [q0, q1) of the tail block in one broadcast each for L and Dinv: 128-wide panels, each an outer panel of its own, lie back to back
static int bcast_tail_panels(gpk_model* m, int q0, int q1, hipStream_t s) {
  const int owner = m->t_own;
  const size_t off = m->poff[q0], cnt = (q1 < m->nP ? m->poff[q1] : m->L_elems) - off;
  const size_t doff = (size_t)(m->pc0[q0] / 128) * 128 * 128, dcnt = (size_t)(q1 - q0) * 128 * 128;
  if (m->virtual_rank) {
    if (!m->vgroup) return gpk_set_err(-1, "internal: virtual rank %d without a group", m->r);
    if (owner == m->r) return 0;
    const gpk_model* o = m->vgroup[owner];  // the caller has made `s` wait for the owner's panel q1 - 1
    GPK_HIP(hipMemcpyAsync(m->dL + off, o->dL + off, cnt * 8, hipMemcpyDeviceToDevice, s));
    GPK_HIP(hipMemcpyAsync(m->dDinv + doff, o->dDinv + doff, dcnt * 8, hipMemcpyDeviceToDevice, s));
    return 0;
  }
  GPK_TRY(comm_bcast(m->dL + off, cnt, owner, s));
  GPK_TRY(comm_bcast(m->dDinv + doff, dcnt, owner, s));
  return 0;
}
// ---- the tail block (gpk_model::ct): hand-over from the cyclic part and the tile pipeline on the tail's owner.  Panel p = p_tail - 1 is the
//      last cyclic panel; its update touches tail columns only.  Owner: three tile columns on the chain stream, the rest on the update
//      stream, then potrf_tail over the tail's panels, each of which is broadcast (communication stream) once all of its rows are in
//      place.  Everybody else: receives the tail's panels in order.
int g_dist_tail_from = -1, g_dist_tail_col = -1, g_dist_tail_owner = -1;
int g_kbuild_split = 0;
static int dist_tail_phase(gpk_model* m) {
  hipStream_t sm = G.s_main, sp = G.s_panel, sc = comm_stream();
  const int nP = m->nP, pt = m->p_tail, p = pt - 1, np_ = m->Npad;
  hipEvent_t* ev_avail = m->ev.data();
  hipEvent_t* ev_trail = m->ev.data() + nP;
  hipEvent_t* ev_bcast = m->ev.data() + 2 * nP + 6;
  // the tail's panels travel dist_tail_chunk at a time (the same chunks on every rank): one collective launch per chunk next to the
  // owner's chain instead of two per panel; a chunk's panels become available together
  const int chunk = std::max(1, G.dist_tail_chunk);
  if (m->r != m->t_own) {
    for (int q0 = pt; q0 < nP; q0 += chunk) {
      const int q1 = std::min(nP, q0 + chunk);
      if (m->virtual_rank) GPK_HIP(hipStreamWaitEvent(sc, m->vgroup[m->t_own]->ev[q1 - 1], 0));
      GPK_TRY(bcast_tail_panels(m, q0, q1, sc));
      for (int q = q0; q < q1; ++q) {
        GPK_HIP(hipEventRecord(ev_bcast[q], sc));
        GPK_HIP(hipEventRecord(ev_avail[q], sc));
      }
    }
    return 0;
  }
  const int w_la = std::min(384, np_ - m->ct);
  if (p > 0) GPK_HIP(hipStreamWaitEvent(sp, ev_trail[p - 1], 0));  // earlier updates of the tail columns ran on s_main
  GPK_HIP(hipStreamWaitEvent(sp, ev_avail[p], 0));
  GPK_TRY(dist_update(m, p, m->ct, m->ct + w_la, sp));
  GPK_HIP(hipStreamWaitEvent(sm, ev_avail[p], 0));
  timer_begin("trailing", sm, dist_update_flops(m, p, m->ct + w_la, np_));
  GPK_TRY(dist_update(m, p, m->ct + w_la, np_, sm));
  timer_end("trailing", sm);
  GPK_HIP(hipEventRecord(ev_trail[p], sm));
  std::function<int(int)> publish = [&](int q) -> int {  // panel q is complete on this rank (potrf_tail: ev[q] recorded on s_main)
    const int q0 = pt + (q - pt) / chunk * chunk, q1 = std::min(nP, q0 + chunk);
    if (q + 1 < q1) return 0;
    GPK_HIP(hipStreamWaitEvent(sc, ev_avail[q], 0));
    GPK_TRY(bcast_tail_panels(m, q0, q1, sc));
    for (int k = q0; k < q1; ++k) GPK_HIP(hipEventRecord(ev_bcast[k], sc));
    return 0;
  };
  return potrf_tail(m, pt, ev_trail[p], &publish);
}
static int dist_la_end(gpk_model* m) {
  // everything after potrf runs on s_main: the last panel must be here and the communication stream drained
  GPK_HIP(hipStreamWaitEvent(G.s_main, m->ev[m->nP - 1], 0));
  GPK_HIP(hipStreamWaitEvent(G.s_main, m->ev[2 * m->nP + 6 + m->nP - 1], 0));
  return 0;
}
// ---- no look-ahead (option lookahead = 0, or fewer than three panels): everything on s_main
static int dist_plain_panel_phase(gpk_model* m, int p) {
  hipStream_t sm = G.s_main;
  if (p == 0) GPK_HIP(hipMemsetAsync(m->dinfo, 0, 16, sm));
  if (panel_owner(m, p) == m->r) GPK_TRY(factor_panel(m, p, sm));
  return bcast_panel(m, p, sm);
}
static int dist_plain_update_phase(gpk_model* m, int p) {
  hipStream_t sm = G.s_main;
  const long from = m->pc0[p] + m->pw[p];
  timer_begin("trailing", sm, dist_update_flops(m, p, from, m->Npad));
  GPK_TRY(dist_update(m, p, from, m->Npad, sm));
  timer_end("trailing", sm);
  return 0;
}

int g_dist_lookahead_used = 0;

// ms[0..n): the rank models of ONE block-cyclic factorisation that live in this process (n = 1: a real rank,
// n = P: every virtual rank, in rank order).  Within a phase the owner of the panel being produced goes first,
// so that in the single-process harness the owner's factor kernels precede the peers' copies in stream order.
int potrf_group(gpk_model* const* ms, int n) {
  if (n < 1 || !ms || !ms[0]) return gpk_set_err(-1, "potrf_group: empty group");
  const int nP = ms[0]->nP, P = ms[0]->P;
  for (int i = 0; i < n; ++i) {
    const gpk_model* m = ms[i];
    if (m->P != P || m->nP != nP || m->Npad != ms[0]->Npad || m->OB != ms[0]->OB || m->ct != ms[0]->ct || m->Pc != ms[0]->Pc || m->t_own != ms[0]->t_own)
      return gpk_set_err(-1, "potrf_group: rank models disagree on the block-cyclic layout");
    if (n > 1 && (!m->virtual_rank || n != P || m->r != i || m->vgroup != ms))
      return gpk_set_err(-1, "potrf_group: a group of %d needs the %d virtual ranks in rank order", n, P);
  }
  auto from_owner = [&](int owner, const std::function<int(gpk_model*)>& f) -> int {
    for (int k = 0; k < n; ++k) GPK_TRY(f(ms[n == 1 ? 0 : (owner + k) % n]));
    return 0;
  };
  g_potrf_lookahead_used = 0;
  const bool la = dist_lookahead(ms[0]);
  g_dist_lookahead_used = la ? 1 : 0;
  if (!la) {
    for (int p = 0; p < nP; ++p) {
      GPK_TRY(from_owner(panel_owner(ms[0], p), [&](gpk_model* m) { return dist_plain_panel_phase(m, p); }));
      GPK_TRY(from_owner(0, [&](gpk_model* m) { return dist_plain_update_phase(m, p); }));
    }
    return 0;
  }
  GPK_TRY(from_owner(0, [&](gpk_model* m) { return dist_la_begin(m); }));
  const int p_tail = ms[0]->ct > 0 ? ms[0]->p_tail : nP;  // from here on: the tail block, one rank's tile pipeline
  g_dist_tail_from = p_tail < nP ? p_tail : -1;
  g_dist_tail_col = p_tail < nP ? ms[0]->ct : -1;
  g_dist_tail_owner = p_tail < nP ? (ms[0]->Pc < P ? P + ms[0]->t_own : ms[0]->t_own) : -1;  // (P + rank: the owner keeps out of the cyclic part)
  for (int p = 0; p < nP; ++p) {
    if (p_tail < nP && p + 1 == p_tail) {
      GPK_TRY(from_owner(ms[0]->t_own, [&](gpk_model* m) { return dist_tail_phase(m); }));
      break;
    }
    GPK_TRY(from_owner(p + 1 < nP ? panel_owner(ms[0], p + 1) : 0, [&](gpk_model* m) { return dist_la_panel_phase(m, p); }));
    GPK_TRY(from_owner(0, [&](gpk_model* m) { return dist_la_update_phase(m, p); }));
  }
  GPK_TRY(from_owner(0, [&](gpk_model* m) { return dist_la_end(m); }));
  return 0;
}

// K(X,X) into this rank's block columns (only tiles on/below the diagonal unless a full copy is kept)
SEScalarForm gpk_model::k_form(int rows_ready) const {
  SEScalarForm f;
  if (dXs && G.se_scalar) { f.scratch = dXs; f.rows_ready = rows_ready; }
  return f;
}
int gpk_model::kxz_form(int Mpad, SEScalarForm* out) {
  *out = SEScalarForm();
  if (!dXs || !G.se_scalar) return 0;
  const size_t need = se_scratch_doubles(Mpad, 0, d, false);
  if (dZs_elems < need) {  // (a buffer from the pool: the same test-set size gets the same pointer back)
    if (dZs) dev_free(dZs, dZs_elems * 8);
    dZs = nullptr;
    dZs_elems = 0;
    GPK_TRY(dev_alloc_t(&dZs, need));
    dZs_elems = need;
  }
  out->scratch = dZs;
  out->cols_scaled = dXs;
  out->cols_pad = Npad;
  return 0;
}
int build_K(gpk_model* m) {
  hipStream_t s = G.s_main;
  const int np_ = m->Npad;
  if (m->P == 1)
    return launch_se_build(m->dX, m->N, np_, m->dX, m->N, np_, m->se, 1, 0, m->dK, m->ldk, m->keep_k ? 0 : 1, s, 0, nullptr, m->k_form());
  // the scaled training points once (also on a rank that owns no column: its Kxz builds read them), every block with rows_ready
  const SEScalarForm kf = m->k_form(1);
  if (kf.scratch) GPK_TRY(launch_se_scale(m->dX, m->N, np_, m->se, m->dX, kf.scratch, s));
  for (const gpk_model::ColRange& cr : m->owned_ranges()) {  // one launch per owned block (cyclic blocks, then the tail block on its owner)
    int nc = m->N - cr.gc0;
    if (nc > cr.ncols) nc = cr.ncols;
    if (nc < 0) nc = 0;
    // column points of this block (clamped pointer: nc == 0 blocks are pure padding)
    const double* Cc = m->dX + (size_t)(nc > 0 ? cr.gc0 : 0) * m->d;
    GPK_TRY(launch_se_build(m->dX, m->N, np_, Cc, nc, cr.ncols, m->se, 1, 0, m->dK + cr.lc0, m->ldk, 1, s, cr.gc0, m->dX, kf));
  }
  return 0;
}

// ------------------------------------------------------------------ triangular solves, single right-hand side
int trsv_forward(gpk_model* m, double* v) {
  hipStream_t s = G.s_main;
  for (int p = 0; p < m->nP; ++p) {
    const int c0 = m->pc0[p], w = m->pw[p], h = m->Npad - c0;
    double* Lp = m->Lpanel(p);
    GPK_TRY(launch_diag_solve(Lp, w, m->pld[p], m->dDinv + (long)(c0 / 128) * 128 * 128, v + c0, 0, s));
    GPK_TRY(launch_gemv_sub(Lp + (long)w * m->pld[p], m->pld[p], h - w, w, v + c0, v + c0 + w, s));
  }
  return 0;
}
int trsv_backward(gpk_model* m, double* v) {
  hipStream_t s = G.s_main;
  for (int p = m->nP - 1; p >= 0; --p) {
    const int c0 = m->pc0[p], w = m->pw[p], h = m->Npad - c0;
    double* Lp = m->Lpanel(p);
    GPK_TRY(launch_gemvT_sub(Lp + (long)w * m->pld[p], m->pld[p], h - w, w, v + c0 + w, v + c0, m->dws, s));
    GPK_TRY(launch_diag_solve(Lp, w, m->pld[p], m->dDinv + (long)(c0 / 128) * 128 * 128, v + c0, 1, s));
  }
  return 0;
}

// Factor m->dK and overwrite v with L^-1 v, the forward substitution riding on the per-panel events of the
// factorisation like the y substitution of fit_device (its two tiny kernels per panel then cost nothing on top of
// the factorisation instead of a latency chain of 2 nP launches after it).  Used by the preference-GP Newton loop.
int potrf_forward(gpk_model* m, double* v) {
  hipStream_t s = G.s_main, sq = G.s_pred;
  if (!(sq && m->P == 1 && G.lookahead && G.fuse_predict && m->nP >= 3)) {
    GPK_TRY(potrf_device(m));
    return trsv_forward(m, v);
  }
  GPK_TRY(model_events(m, 3 * m->nP + 6));
  hipEvent_t ev_in = m->ev[2 * m->nP + 2];
  GPK_HIP(hipEventRecord(ev_in, s));  // v and dK are produced on s_main
  GPK_HIP(hipStreamWaitEvent(sq, ev_in, 0));
  GPK_TRY(potrf_device(m));
  if (!g_potrf_lookahead_used) return trsv_forward(m, v);
  hipEvent_t* ev_panel = m->ev.data();
  const int np_ = m->Npad;
  for (int p = 0; p < m->nP; ++p) {
    const int c0 = m->pc0[p], w = m->pw[p], h = np_ - c0;
    double* Lp = m->Lpanel(p);
    GPK_HIP(hipStreamWaitEvent(sq, ev_panel[p], 0));
    GPK_TRY(launch_diag_solve(Lp, w, m->pld[p], m->dDinv + (long)(c0 / 128) * 128 * 128, v + c0, 0, sq));
    GPK_TRY(launch_gemv_sub(Lp + (long)w * m->pld[p], m->pld[p], h - w, w, v + c0, v + c0 + w, sq));
  }
  hipEvent_t ev_out = m->ev[2 * m->nP + 3];
  GPK_HIP(hipEventRecord(ev_out, sq));
  GPK_HIP(hipStreamWaitEvent(s, ev_out, 0));
  return 0;
}

// ------------------------------------------------------------------ fused fit (inputs already in HBM)
static int predict_diag(gpk_model* m, int p, int Mpad, double* dR, double* dV, hipStream_t s);
static int predict_update(gpk_model* m, int p, int Mpad, int cg0, int ncols, double* dR, double* dV, hipStream_t s);
static int predict_outer_update(gpk_model* m, int o, int Mpad, int cg0, int ncols, double* dR, double* dV, hipStream_t s);
struct PredDefer;
static std::shared_ptr<PredDefer> predict_defer_make(const gpk_model* m);
static int predict_step(gpk_model* m, int p, int Mpad, double* dR, double* dV, hipStream_t s, const PredDefer* df);

// Block-cyclic fit of the rank models living in this process (see potrf_group): every rank builds its block columns
// of K, the group factors, every rank then holds all of L and runs the y substitution redundantly.
// gp (optional): a sharded prediction fused into the fit -- rank r solves the test points [r Mc, (r+1) Mc) into gp->bufs[i].
// Where the look-ahead schedule runs and the process has a substitution stream, the forward substitutions (y on every rank, the
// rank's rows of Kzx) are queued on it behind the per-panel "available on this rank" events of the factorisation (the owner: its
// own factorisation, everybody else: the broadcast), as fit_device does on one GPU: they fill the chain- and link-bound stretches
// of the block-cyclic factorisation instead of following it (2 nP dependent launches for y alone: 1.3 ms at N = 16384).
struct GroupPredict {
  const double* dZ = nullptr;
  int M = 0, Mc = 0, Mpad = 0;          // all test points, per rank, per rank padded
  struct Bufs { double *dR = nullptr, *dV = nullptr, *dall = nullptr; };
  std::vector<Bufs> bufs;               // one per local model
};
int g_dist_pipelined = 0;  // 1 if the last block-cyclic fit queued its substitutions behind the panel events
static int fit_group_impl(gpk_model* const* ms, int n, const GroupPredict* gp) {
  hipStream_t s = G.s_main;
  for (int i = 0; i < n; ++i) {
    gpk_model* m = ms[i];
    timer_begin("kbuild", s, 4.0 * m->Npad * (double)m->ldk + 8.0 * m->N * m->d);
    GPK_TRY(build_K(m));
    timer_end("kbuild", s);
  }
  timer_begin("potrf", s);
  GPK_TRY(potrf_group(ms, n));
  timer_end("potrf", s);
  hipStream_t sq = G.s_pred;
  const bool pipelined = sq && G.fuse_predict && G.dist_pipeline && g_dist_lookahead_used;  // (virtual ranks: the stand-in broadcasts share this stream -- queued first, so the substitutions follow them)
  g_dist_pipelined = pipelined ? 1 : 0;
  for (int i = 0; i < n; ++i) {
    gpk_model* m = ms[i];
    const int np_ = m->Npad;
    GPK_TRY(launch_sum_log_diag(m->dDinv, m->N, m->dscal + 0, s));
    const int m0 = gp ? ((m->P > 1) ? m->r * gp->Mc : 0) : 0;
    const int cnt = gp ? std::max(0, std::min(gp->Mc, gp->M - m0)) : 0;
    const int mp = round_up(std::max(cnt, 1), 128);
    if (gp) GPK_HIP(hipMemsetAsync(m->shard_loc, 0, (size_t)2 * gp->Mpad * 8, s));
    if (pipelined && (m->has_y || cnt > 0)) {
      // everything below only needs panels <= p on THIS rank: m->ev[p] (dist_publish_panel); m->ev[2 nP] = the K build is done
      hipEvent_t* ev_avail = m->ev.data();
      GPK_HIP(hipStreamWaitEvent(sq, m->ev[2 * m->nP], 0));
      if (m->has_y) GPK_HIP(hipMemcpyAsync(m->dw, m->dy, (size_t)np_ * 8, hipMemcpyDeviceToDevice, sq));
      if (cnt > 0) {
        SEScalarForm zf;
        GPK_TRY(m->kxz_form(mp, &zf));
        GPK_TRY(launch_se_build(gp->dZ + (size_t)m0 * m->d, cnt, mp, m->dX, m->N, np_, m->se, 0, 0, gp->bufs[i].dR, np_, 0, sq, 0, nullptr, zf));
      }
      for (int p = 0; p < m->nP; ++p) {
        const int c0 = m->pc0[p], w = m->pw[p], h = np_ - c0;
        GPK_HIP(hipStreamWaitEvent(sq, ev_avail[p], 0));
        if (m->has_y) {
          double* Lp = m->Lpanel(p);
          GPK_TRY(launch_diag_solve(Lp, w, m->pld[p], m->dDinv + (long)(c0 / 128) * 128 * 128, m->dw + c0, 0, sq));
          GPK_TRY(launch_gemv_sub(Lp + (long)w * m->pld[p], m->pld[p], h - w, w, m->dw + c0, m->dw + c0 + w, sq));
        }
        if (cnt > 0) GPK_TRY(predict_step(m, p, mp, gp->bufs[i].dR, gp->bufs[i].dV, sq, nullptr));
      }
      hipEvent_t ev_pd = m->ev[2 * m->nP + 3];
      GPK_HIP(hipEventRecord(ev_pd, sq));
      GPK_HIP(hipStreamWaitEvent(s, ev_pd, 0));
      if (m->has_y) GPK_TRY(launch_dot(m->dw, m->dw, np_, m->dscal + 1, s));
      if (cnt > 0) GPK_TRY(launch_row_stats(gp->bufs[i].dV, np_, cnt, np_, m->dw, m->se.sf2, m->shard_loc, m->shard_loc + gp->Mpad, s));
      continue;
    }
    if (m->has_y) {
      timer_begin("solve_y", s);
      GPK_HIP(hipMemcpyAsync(m->dw, m->dy, (size_t)np_ * 8, hipMemcpyDeviceToDevice, s));
      GPK_TRY(trsv_forward(m, m->dw));
      GPK_TRY(launch_dot(m->dw, m->dw, np_, m->dscal + 1, s));
      timer_end("solve_y", s);
    }
    if (cnt > 0) GPK_TRY(predict_device(m, gp->dZ + (size_t)m0 * m->d, cnt, mp, gp->bufs[i].dR, gp->bufs[i].dV, m->shard_loc, m->shard_loc + gp->Mpad));
  }
  std::vector<double> hs((size_t)2 * n, 0.0);
  std::vector<int> hinfo(n, 0);
  for (int i = 0; i < n; ++i) {
    GPK_HIP(hipMemcpyAsync(&hs[2 * i], ms[i]->dscal, 16, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipMemcpyAsync(&hinfo[i], ms[i]->dinfo, 4, hipMemcpyDeviceToHost, s));
  }
  GPK_HIP(hipStreamSynchronize(s));
  // Only the owner of the failing panel saw the pivot, and once one tile has failed the owners of later panels fail as
  // well (on what the identity factors of the failed tile left them): the verdict is the FIRST failing minor over all
  // ranks, as LAPACK reports it -- the smallest positive info, found with the max-reduction as the largest 2^30 - info.
  int worst = 0;
  for (int i = 0; i < n; ++i) {
    GPK_TRY(info_timeout_rc(hinfo[i]));
    if (hinfo[i] > 0 && (worst == 0 || hinfo[i] < worst)) worst = hinfo[i];
  }
  if (!ms[0]->virtual_rank && ms[0]->P > 1) {
    double hi = worst > 0 ? (double)((1 << 30) - worst) : 0.0;
    GPK_TRY(gpk_comm_allreduce_max(&hi, 1));
    worst = hi > 0.0 ? (1 << 30) - (int)hi : 0;
  }
  for (int i = 0; i < n; ++i) {
    gpk_model* m = ms[i];
    m->info = worst;
    if (worst > 0) continue;
    m->factored = true;
    m->logdet = hs[2 * i];
    m->quad = m->has_y ? hs[2 * i + 1] : 0.0;
    m->nlml = m->has_y ? 0.5 * m->quad + m->logdet + 0.5 * m->N * std::log(2.0 * M_PI) : m->logdet;  // GPr.py:64-69
  }
  if (worst > 0) return worst > ms[0]->N ? ms[0]->N : worst;
  return 0;
}
int fit_group(gpk_model* const* ms, int n) { return fit_group_impl(ms, n, nullptr); }

// Everything one fit queues on the three streams, from the K build to the read-back into the model's pinned staging
// buffer -- no synchronisation, no allocation of device memory.
static int fit_enqueue(gpk_model* m, const FusedPredict* fp) {
  hipStream_t s = G.s_main;
  const int np_ = m->Npad;
  if (G.stamps && G.dstamps) {
    GPK_HIP(hipMemsetAsync(G.dstamps, 0, 1024 * 8, s));
    GPK_TRY(launch_stamp(G.dstamps + 447, s));  // "stamp_start": the fit begins (before the K build)
  }
  timer_begin("kbuild", s, (m->keep_k ? 8.0 : 4.0) * np_ * np_ + 8.0 * m->N * m->d);
  // K in two launches where the look-ahead schedule follows (option kbuild_split, off: measured, no gain): block column 0 over
  // all rows, then the lower triangle behind it -- the chain stream factors panel 0 (0.4 ms of leaves and skinny GEMMs at
  // N = 16384) beside the second launch instead of behind it.  The same tiles from the same kernels: the same bits.
  const int w0 = m->ow[0];
  const bool split = G.kbuild_split && !m->keep_k && !m->k_only && m->P == 1 && G.lookahead && m->nP >= 3 && tail_first_panel(m) > 0 &&
                     np_ >= 4 * w0 && m->N > w0;
  g_kbuild_split = split ? 1 : 0;
  m->ev_kfirst = nullptr;
  if (split) {
    GPK_TRY(model_events(m, 3 * m->nP + 6));
    GPK_TRY(launch_se_build(m->dX, m->N, np_, m->dX, w0, w0, m->se, 1, 0, m->dK, m->ldk, 1, s, 0, m->dX, m->k_form()));
    m->ev_kfirst = m->ev[2 * m->nP + 5];
    GPK_HIP(hipEventRecord(m->ev_kfirst, s));
    SEScalarForm fb = m->k_form(1);  // rows w0.. of the scaled set the first launch has left
    if (fb.scratch) {
      fb.scratch += (size_t)w0 * m->d;
      fb.rows_norms = m->dXs + (size_t)np_ * m->d + w0;
    }
    const double* Xb = m->dX + (size_t)w0 * m->d;
    GPK_TRY(launch_se_build(Xb, m->N - w0, np_ - w0, Xb, m->N - w0, np_ - w0, m->se, 1, 0, m->dK + (size_t)w0 * m->ldk + w0, m->ldk, 1, s, 0, m->dX, fb));
  } else {
    GPK_TRY(launch_se_build(m->dX, m->N, np_, m->dX, m->N, np_, m->se, 1, 0, m->dK, m->ldk, m->keep_k ? 0 : 1, s, 0, nullptr, m->k_form()));
  }
  timer_end("kbuild", s);
  if (m->keep_k)
    GPK_HIP(hipMemcpy2DAsync(m->dKcopy, (size_t)np_ * 8, m->dK, (size_t)m->ldk * 8, (size_t)np_ * 8, np_, hipMemcpyDeviceToDevice, s));
  if (m->k_only) return 0;
  timer_begin("potrf", s);
  hipEvent_t ev_kb = nullptr;
  const bool fuse = fp && G.fuse_predict && G.s_pred && m->P == 1 && G.lookahead && m->nP >= 3;
  if (G.fuse_predict && G.s_pred && m->P == 1 && G.lookahead && m->nP >= 3 && (fuse || m->has_y)) {
    // s_pred may start once K is built (Kzx only needs X, the y substitution only panels and y)
    GPK_TRY(model_events(m, 3 * m->nP + 6));
    ev_kb = m->ev[2 * m->nP + 2];
    GPK_HIP(hipEventRecord(ev_kb, s));
  }
  // Everything that only needs panels <= p goes to s_pred behind the per-panel events of the
  // factorisation: the forward substitution w = L^-1 y (two tiny kernels per panel) and, for a fused
  // fit+predict, the forward substitution of the prediction.  While the trailing updates saturate the
  // chip this merely re-orders work; in the latency-bound tail of the factorisation it fills idle CUs.
  // (The host queues them after the whole factorisation; queueing panel p's right after panel p was measured: no gain.)
  const bool pipelined = G.s_pred && m->P == 1 && G.fuse_predict && G.lookahead && m->nP >= 3 && (m->has_y || fuse);
  bool fused_done = false;
  hipStream_t sq = G.s_pred;
  hipStream_t sy = sq;  // one stream for both substitutions (see gpk_init on the stream count)
  std::shared_ptr<PredDefer> defer = fuse ? predict_defer_make(m) : nullptr;
  std::function<int(int)> hook = [&](int p) -> int {
    hipEvent_t* ev_panel = m->ev.data();
    const int c0 = m->pc0[p], w = m->pw[p], h = np_ - c0;
    if (m->has_y) {
      double* Lp = m->Lpanel(p);
      GPK_HIP(hipStreamWaitEvent(sy, ev_panel[p], 0));
      GPK_TRY(launch_diag_solve(Lp, w, m->pld[p], m->dDinv + (long)(c0 / 128) * 128 * 128, m->dw + c0, 0, sy));
      GPK_TRY(launch_gemv_sub(Lp + (long)w * m->pld[p], m->pld[p], h - w, w, m->dw + c0, m->dw + c0 + w, sy));
    }
    if (fuse) {
      if (!m->has_y || sy != sq) GPK_HIP(hipStreamWaitEvent(sq, ev_panel[p], 0));
      GPK_TRY(predict_step(m, p, fp->Mpad, fp->dR, fp->dV, sq, defer.get()));
    }
    if (G.stamps && G.dstamps) GPK_TRY(launch_stamp(G.dstamps + 512 + p, sq));
    return 0;
  };
  if (pipelined) {
    if (!ev_kb) return gpk_set_err(-1, "internal: pipeline start event missing");
    GPK_HIP(hipStreamWaitEvent(sq, ev_kb, 0));
    if (sy != sq) GPK_HIP(hipStreamWaitEvent(sy, ev_kb, 0));
    if (m->has_y) GPK_HIP(hipMemcpyAsync(m->dw, m->dy, (size_t)np_ * 8, hipMemcpyDeviceToDevice, sy));
    if (fuse) {
      timer_begin("predict", sq);
      SEScalarForm zf;
      GPK_TRY(m->kxz_form(fp->Mpad, &zf));
      GPK_TRY(launch_se_build(fp->dZ, fp->M, fp->Mpad, m->dX, m->N, np_, m->se, 0, 0, fp->dR, np_, 0, sq, 0, nullptr, zf));
    }
  }
  const auto host_t0 = std::chrono::steady_clock::now();
  const int prc = potrf_device(m);
  m->ev_kfirst = nullptr;
  GPK_TRY(prc);
  G.last_potrf_enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
  timer_end("potrf", s);
  // s_main has joined the last panel: the log-determinant can run while the substitution stream finishes
  GPK_TRY(launch_sum_log_diag(m->dDinv, m->N, m->dscal + 0, s));
  const auto host_t1 = std::chrono::steady_clock::now();
  if (pipelined) {
    if (!g_potrf_lookahead_used) return gpk_set_err(-1, "internal: pipelined substitutions without look-ahead");
    for (int p = 0; p < m->nP; ++p) GPK_TRY(hook(p));
    if (fuse) {
      timer_end("predict", sq);
      fused_done = true;
    }
    hipEvent_t ev_pd = m->ev[2 * m->nP + 3];
    GPK_HIP(hipEventRecord(ev_pd, sq));
    GPK_HIP(hipStreamWaitEvent(s, ev_pd, 0));
    if (sy != sq) {
      hipEvent_t ev_yd = m->ev[2 * m->nP + 4];
      GPK_HIP(hipEventRecord(ev_yd, sy));
      GPK_HIP(hipStreamWaitEvent(s, ev_yd, 0));
    }
  }
  G.last_pipe_enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t1).count();
  if (m->has_y) {
    timer_begin("solve_y", s);
    if (!pipelined) {
      GPK_HIP(hipMemcpyAsync(m->dw, m->dy, (size_t)np_ * 8, hipMemcpyDeviceToDevice, s));
      GPK_TRY(trsv_forward(m, m->dw));
    }
    GPK_TRY(launch_dot(m->dw, m->dw, np_, m->dscal + 1, s));
    timer_end("solve_y", s);
  }
  if (fp) {
    if (fused_done) {
      GPK_TRY(launch_row_stats(fp->dV, np_, fp->M, np_, m->dw, m->se.sf2, fp->dmean, fp->dvar, s));
    } else {
      GPK_TRY(predict_device(m, fp->dZ, fp->M, fp->Mpad, fp->dR, fp->dV, fp->dmean, fp->dvar));
    }
  }
  // read-back into the model's pinned staging buffer: one synchronisation (the caller's) for scalars, info, mean, variance
  GPK_HIP(hipMemcpyAsync(m->hpin, m->dscal, 16, hipMemcpyDeviceToHost, s));
  GPK_HIP(hipMemcpyAsync(m->hpin + 2, m->dinfo, 4, hipMemcpyDeviceToHost, s));
  if (fp) {
    GPK_HIP(hipMemcpyAsync(m->hpin + 4, fp->dmean, (size_t)fp->M * 8, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipMemcpyAsync(m->hpin + 4 + fp->M, fp->dvar, (size_t)fp->M * 8, hipMemcpyDeviceToHost, s));
  }
  if (G.stamps && G.dstamps) GPK_HIP(hipMemcpyAsync(G.hstamps.data(), G.dstamps, 1024 * 8, hipMemcpyDeviceToHost, s));
  return 0;
}

int fit_device(gpk_model* m, const FusedPredict* fp) {
  hipStream_t s = G.s_main;
  if (m->P > 1) {
    if (fp) return gpk_set_err(-1, "internal: fused prediction on a block-cyclic model");
    if (m->virtual_rank) return gpk_set_err(-1, "internal: virtual ranks are fitted as a group (fit_group)");
    gpk_model* one = m;
    return fit_group(&one, 1);
  }
  if (m->k_only && !m->keep_k) return gpk_set_err(-1, "internal: a K-only model must keep K");
  // the pinned read-back buffer (pooled by size: a destroyed model's buffer serves the next model of that shape)
  const size_t need = 4 + (fp ? 2 * (size_t)fp->M : 0);
  if (m->hpin_doubles < need) {
    if (m->hpin) G.hpin_pool[m->hpin_doubles].push_back(m->hpin);
    m->hpin = nullptr;
    auto it = G.hpin_pool.find(need);
    if (it != G.hpin_pool.end() && !it->second.empty()) {
      m->hpin = it->second.back();
      it->second.pop_back();
    } else {
      GPK_HIP(hipHostMalloc((void**)&m->hpin, need * 8));
    }
    m->hpin_doubles = need;
  }
  if (G.stamps && !G.dstamps && m->nP <= 447) {
    GPK_HIP(hipMalloc((void**)&G.dstamps, 1024 * 8));
    G.hstamps.assign(1024, 0);
  }
  GPK_TRY(model_events(m, 10 * m->nP + 8));
  if (corner_boundary(m)) GPK_TRY(corner_stream_ensure());
  m->hpin[2] = 0.0;
  int rc = fit_enqueue(m, fp);
  if (rc) {
    (void)hipStreamSynchronize(s);
    return rc;
  }
  GPK_HIP(hipStreamSynchronize(s));
  if (m->k_only) return 0;
  int hinfo = 0;
  std::memcpy(&hinfo, m->hpin + 2, 4);
  if (fp && fp->h_mean) std::memcpy(fp->h_mean, m->hpin + 4, (size_t)fp->M * 8);
  if (fp && fp->h_var) std::memcpy(fp->h_var, m->hpin + 4 + fp->M, (size_t)fp->M * 8);
  m->info = hinfo;
  GPK_TRY(info_timeout_rc(hinfo));
  if (hinfo > 0) return hinfo > m->N ? m->N : hinfo;
  m->factored = true;
  m->logdet = m->hpin[0];
  m->quad = m->has_y ? m->hpin[1] : 0.0;
  // GPr.py:64-69: err_y + det_Kxx + occams_razor
  m->nlml = m->has_y ? 0.5 * m->quad + m->logdet + 0.5 * m->N * std::log(2.0 * M_PI) : m->logdet;
  return 0;
}

// ------------------------------------------------------------------ prediction (GPr.py:45-54)
// V = Kzx L^-T by blocked forward substitution on the ROWS of Kzx (the same NT GEMM as the
// factorisation), then mean_j = V_j . w and var_j = sf2 - |V_j|^2 in one pass.
// forward substitution of panel p's diagonal block on the rows of R:  V[:, panel p] = R[:, panel p] * Lpp^-T
static int predict_diag(gpk_model* m, int p, int Mpad, double* dR, double* dV, hipStream_t s) {
  const int np_ = m->Npad;
  const int c0 = m->pc0[p], w = m->pw[p];
  const long ld = m->pld[p];
  double* Lp = m->Lpanel(p);
  for (int sb = 0; sb < w / 128; ++sb) {
    const int cs = c0 + sb * 128, ro = cs - c0;
    if (sb > 0) {
      GemmNT g{};
      g.A = dV + c0; g.lda = np_;
      g.B = Lp + (long)ro * ld; g.ldb = ld;
      g.C = dR + cs; g.ldc = np_;
      g.m = Mpad; g.n = 128; g.k = ro; g.sub = 1;
      GPK_TRY(launch_gemm_nt(g, false, s));
    }
    GemmNT g{};
    g.A = dR + cs; g.lda = np_;
    g.B = m->dDinv + (long)(cs / 128) * 128 * 128; g.ldb = 128;
    g.C = dV + cs; g.ldc = np_;
    g.m = Mpad; g.n = 128; g.k = 128; g.sub = 0;
    GPK_TRY(launch_gemm_nt(g, false, s));
  }
  return 0;
}
// R[:, cg0 : cg0+ncols] -= V[:, panel p] * Lp[(cg0-c0) : +ncols, :]^T
static int predict_update(gpk_model* m, int p, int Mpad, int cg0, int ncols, double* dR, double* dV, hipStream_t s) {
  if (ncols <= 0) return 0;
  const int np_ = m->Npad;
  const int c0 = m->pc0[p], w = m->pw[p];
  const long ld = m->pld[p];
  GemmNT g{};
  g.A = dV + c0; g.lda = np_;
  g.B = m->Lpanel(p) + (long)(cg0 - c0) * ld; g.ldb = ld;
  g.C = dR + cg0; g.ldc = np_;
  g.m = Mpad; g.n = ncols; g.k = w; g.sub = 1;
  // few 128x128 tiles (the look-ahead block column): 64x64 tiles put 4x as many CUs on the critical path
  if ((long)(Mpad / 128) * (ncols / 128) < G.pred_small_below()) return launch_gemm_nt(g, false, s);
  return launch_gemm_auto(g, s);
}
// the same with a whole outer panel: R[:, cg0 : cg0+ncols] -= V[:, outer o] * Lo[(cg0-oc0) : +ncols, :]^T  (K = ow[o])
static int predict_outer_update(gpk_model* m, int o, int Mpad, int cg0, int ncols, double* dR, double* dV, hipStream_t s) {
  if (ncols <= 0) return 0;
  const int np_ = m->Npad;
  const int c0 = m->oc0[o], W = m->ow[o];
  GemmNT g{};
  g.A = dV + c0; g.lda = np_;
  g.B = m->Lo(o) + (long)(cg0 - c0) * W; g.ldb = W;
  g.C = dR + cg0; g.ldc = np_;
  g.m = Mpad; g.n = ncols; g.k = W; g.sub = 1;
  if ((long)(Mpad / 128) * (ncols / 128) < G.pred_small_below()) return launch_gemm_nt(g, false, s);
  return launch_gemm_auto(g, s);
}
// Everything inner panel p contributes to the row substitution, in the two-level order: its diagonal solve, its
// update of the remaining columns of ITS OUTER block column (K = inner width) and, when p closes the outer panel,
// the outer panel's update of all columns to the right (K = outer width).
// Deferred far updates (option pred_defer_from): with b the first outer-panel boundary at or after the chosen column, an outer
// panel that ends at or before b updates right away only the columns below b; the columns from b on get the updates of ALL
// those panels, in panel order, when the substitution reaches b -- the same updates of every element in the same order, hence
// the same bits, only later: in the factorisation's chain-bound last third, where CUs idle, instead of beside the bulk updates
// that bound the first two (N = 16384: the first 15 panel steps take 22.3 instead of 23.7 ms, the posterior 33.1 instead of
// 33.9; several boundaries instead of one were measured and lose: the later ones run into the end of the factorisation).
struct PredDefer {
  int b = 0;  // the boundary column
};
static int predict_defer_col(const gpk_model* m) {
  int want = G.pred_defer_from;
  if (want < 0) want = (m->Npad >= 12288) ? (int)((long)m->Npad * 5 / 8) : 0;  // automatic: where the N = 16384 schedule turns chain-bound
  if (want <= 0) return 0;
  for (int o = 1; o < m->nO; ++o)
    if (m->oc0[o] >= want) return m->oc0[o];
  return 0;
}
static int predict_step(gpk_model* m, int p, int Mpad, double* dR, double* dV, hipStream_t s, const PredDefer* df = nullptr) {
  const int o = m->pouter[p];
  const int cend = m->oc0[o] + m->ow[o], c1 = m->pc0[p] + m->pw[p];
  if (df && p == m->ofirst[o] && m->oc0[o] == df->b)
    for (int q = 0; q < o; ++q) GPK_TRY(predict_outer_update(m, q, Mpad, df->b, m->Npad - df->b, dR, dV, s));
  GPK_TRY(predict_diag(m, p, Mpad, dR, dV, s));
  GPK_TRY(predict_update(m, p, Mpad, c1, cend - c1, dR, dV, s));
  if (p == m->oend[o] - 1) {
    const int upto = (df && cend <= df->b) ? df->b : m->Npad;
    GPK_TRY(predict_outer_update(m, o, Mpad, cend, upto - cend, dR, dV, s));
  }
  return 0;
}

int g_pred_defer_col = 0;  // boundary the last fused prediction deferred its far updates to (0: none)
static std::shared_ptr<PredDefer> predict_defer_make(const gpk_model* m) {
  const int b = predict_defer_col(m);
  g_pred_defer_col = b > 0 ? b : 0;
  if (b <= 0) return nullptr;
  auto d = std::make_shared<PredDefer>();
  d->b = b;
  return d;
}

// V = R L^-T for Mpad rows (multiple of 128) of width Npad, R destroyed; s_main carries the result.
int rows_forward_solve(gpk_model* m, int Mpad, double* dR, double* dV) {
  hipStream_t sm = G.s_main, sp = G.s_panel;
  const int np_ = m->Npad, nP = m->nP, nO = m->nO;
  if (!G.lookahead || nP < 3) {
    for (int p = 0; p < nP; ++p) GPK_TRY(predict_step(m, p, Mpad, dR, dV, sm, nullptr));
    return 0;
  }
  // one outer panel of look-ahead, as in the factorisation: the latency-bound diagonal solves (and inner updates)
  // of outer block column o+1 run on s_panel while the wide update of outer panel o runs on s_main.
  auto outer_diag = [&](int o, hipStream_t s) -> int {
    const int cend = m->oc0[o] + m->ow[o];
    for (int p = m->ofirst[o]; p < m->oend[o]; ++p) {
      const int c1 = m->pc0[p] + m->pw[p];
      GPK_TRY(predict_diag(m, p, Mpad, dR, dV, s));
      GPK_TRY(predict_update(m, p, Mpad, c1, cend - c1, dR, dV, s));
    }
    return 0;
  };
  GPK_TRY(model_events(m, 3 * nP + 6));
  hipEvent_t* ev_diag = m->ev.data();
  hipEvent_t* ev_upd = m->ev.data() + nP;
  hipEvent_t ev_start = m->ev[2 * nP];
  GPK_HIP(hipEventRecord(ev_start, sm));
  GPK_HIP(hipStreamWaitEvent(sp, ev_start, 0));
  GPK_TRY(outer_diag(0, sp));
  GPK_HIP(hipEventRecord(ev_diag[0], sp));
  for (int o = 0; o < nO; ++o) {
    const int c1 = m->oc0[o] + m->ow[o];
    const int wn = (o + 1 < nO) ? m->ow[o + 1] : 0;
    if (o + 1 < nO) {
      if (o > 0) GPK_HIP(hipStreamWaitEvent(sp, ev_upd[o - 1], 0));
      GPK_TRY(predict_outer_update(m, o, Mpad, c1, wn, dR, dV, sp));
      GPK_TRY(outer_diag(o + 1, sp));
      GPK_HIP(hipEventRecord(ev_diag[o + 1], sp));
    }
    GPK_HIP(hipStreamWaitEvent(sm, ev_diag[o], 0));
    GPK_TRY(predict_outer_update(m, o, Mpad, c1 + wn, np_ - c1 - wn, dR, dV, sm));
    GPK_HIP(hipEventRecord(ev_upd[o], sm));
  }
  GPK_HIP(hipStreamWaitEvent(sm, ev_diag[nO - 1], 0));
  return 0;
}

// Y = R L^-1 for Mpad rows (the mirror of rows_forward_solve; together they apply K^-1 = L^-T L^-1 to many
// right-hand sides on the MFMA GEMM: gpk_solve, the replacement of the explicit iK of GPr.py:63 / GPpref.py:129).
// The products are R . L ("NN"), the kernel is NT, so the caller supplies U = L^T as a dense Npad x Npad array
// (only U[j, k >= block of j] is read; rows_backward_prepare fills it panel by panel) and the transposed inverted
// diagonal tiles DinvT (128 x nT*128: DinvT[j][t*128 + k] = Dinv_t[k][j]).  Right-looking from the last panel:
//   Y[:, p] = R[:, p] Lpp^-1 (sub-blocks last to first),   R[:, 0:c0] -= Y[:, p] U[0:c0, p]^T   (one GEMM, K = w).
int rows_backward_prepare(gpk_model* m, double* dU, double* dDinvT, hipStream_t s) {
  const int np_ = m->Npad;
  for (int p = 0; p < m->nP; ++p) {
    const int c0 = m->pc0[p], w = m->pw[p];
    GPK_TRY(launch_transpose(m->Lpanel(p), m->pld[p], np_ - c0, w, dU + (long)c0 * np_ + c0, np_, s));
  }
  return launch_transpose(m->dDinv, 128, m->nT * 128, 128, dDinvT, (long)m->nT * 128, s);
}
int rows_backward_solve(gpk_model* m, int Mpad, double* dR, double* dY, const double* dU, const double* dDinvT) {
  hipStream_t s = G.s_main;
  const int np_ = m->Npad;
  const long ldd = (long)m->nT * 128;
  for (int p = m->nP - 1; p >= 0; --p) {
    const int c0 = m->pc0[p], w = m->pw[p];
    for (int sb = w / 128 - 1; sb >= 0; --sb) {
      const int cs = c0 + sb * 128;
      const int after = c0 + w - cs - 128;  // columns of this panel right of the sub-block, already solved
      if (after > 0) {
        GemmNT g{};
        g.A = dY + cs + 128; g.lda = np_;
        g.B = dU + (long)cs * np_ + cs + 128; g.ldb = np_;
        g.C = dR + cs; g.ldc = np_;
        g.m = Mpad; g.n = 128; g.k = after; g.sub = 1;
        GPK_TRY(launch_gemm_nt(g, false, s));
      }
      GemmNT g{};
      g.A = dR + cs; g.lda = np_;
      g.B = dDinvT + (long)(cs / 128) * 128; g.ldb = ldd;
      g.C = dY + cs; g.ldc = np_;
      g.m = Mpad; g.n = 128; g.k = 128; g.sub = 0;
      GPK_TRY(launch_gemm_nt(g, false, s));
    }
    if (c0 > 0) {
      GemmNT g{};
      g.A = dY + c0; g.lda = np_;
      g.B = dU + c0; g.ldb = np_;
      g.C = dR; g.ldc = np_;
      g.m = Mpad; g.n = c0; g.k = w; g.sub = 1;
      if ((long)(Mpad / 128) * (c0 / 128) < G.pred_small_below()) GPK_TRY(launch_gemm_nt(g, false, s));
      else GPK_TRY(launch_gemm_auto(g, s));
    }
  }
  return 0;
}

int predict_device(gpk_model* m, const double* dZ, int M, int Mpad, double* dR, double* dV, double* dmean,
                   double* dvar) {
  hipStream_t sm = G.s_main;
  const int np_ = m->Npad;
  timer_begin("predict", sm);
  SEScalarForm zf;
  GPK_TRY(m->kxz_form(Mpad, &zf));
  GPK_TRY(launch_se_build(dZ, M, Mpad, m->dX, m->N, np_, m->se, 0, 0, dR, np_, 0, sm, 0, nullptr, zf));
  GPK_TRY(rows_forward_solve(m, Mpad, dR, dV));
  GPK_TRY(launch_row_stats(dV, np_, M, np_, m->dw, m->se.sf2, dmean, dvar, sm));
  timer_end("predict", sm);
  return 0;
}

// ------------------------------------------------------------------ gradient of the nlml (SURVEY 8f "next": hyper-parameter fit)
// d nlml / d theta = 1/2 tr((Kinv - alpha alpha^T) dK/dtheta),  theta = log_hyp (GPr.py:93-97 parametrisation).
// Kinv comes from the factor already in HBM with two sweeps of the NT GEMM, both at the N^3/3 flop count of the
// factorisation because they only touch the non-zero (upper) part of V = L^-T:
//   1. V = I L^-T   : the row substitution of predict_device applied to the identity; after panel p only rows
//                     < c0+w of V[:, panel p] are non-zero, so every GEMM runs on the first c0+w rows only.
//   2. -Kinv = -V V^T (lower): one rank-w update per panel on the leading (c0+w)^2 block.
// The d+2 traces are then one fused pass over the lower triangle that recomputes the SE kernel on the fly.
int nlml_grad_device(gpk_model* m, double* h_grad) {
  if (m->P > 1) return gpk_set_err(-1, "gpk_nlml_grad: not available on a block-cyclic (multi-GPU) model");
  if (!m->has_y) return gpk_set_err(-1, "gpk_nlml_grad: model was fitted without targets");
  hipStream_t s = G.s_main;
  const int np_ = m->Npad, N = m->N, d = m->d;
  const size_t sq = (size_t)np_ * np_;
  const int nt = (N + 127) / 128;
  double *dR = nullptr, *dV = nullptr, *dNK = nullptr, *dalpha = nullptr, *dpart = nullptr, *dout = nullptr;
  int rc = 0;
  rc |= dev_alloc_t(&dR, sq);
  rc |= dev_alloc_t(&dV, sq);
  rc |= dev_alloc_t(&dNK, sq);
  rc |= dev_alloc_t(&dalpha, (size_t)np_);
  rc |= dev_alloc_t(&dpart, (size_t)nt * nt * (d + 2));
  rc |= dev_alloc_t(&dout, (size_t)(d + 2));
  auto cleanup = [&]() {
    if (dR) dev_free(dR, sq * 8);
    if (dV) dev_free(dV, sq * 8);
    if (dNK) dev_free(dNK, sq * 8);
    if (dalpha) dev_free(dalpha, (size_t)np_ * 8);
    if (dpart) dev_free(dpart, (size_t)nt * nt * (d + 2) * 8);
    if (dout) dev_free(dout, (size_t)(d + 2) * 8);
  };
  if (rc) { cleanup(); return rc; }
  auto body = [&]() -> int {
    timer_begin("grad_inv", s, 2.0 * (double)np_ * np_ * np_ / 3.0);
    GPK_HIP(hipMemsetAsync(dR, 0, sq * 8, s));
    GPK_HIP(hipMemsetAsync(dV, 0, sq * 8, s));
    GPK_HIP(hipMemsetAsync(dNK, 0, sq * 8, s));
    GPK_TRY(launch_set_diag(dR, np_, np_, 1.0, s));
    // both sweeps by OUTER panels (K = outer width for the wide updates); rows restricted to the non-zero part
    for (int o = 0; o < m->nO; ++o) {
      const int cend = m->oc0[o] + m->ow[o];
      for (int p = m->ofirst[o]; p < m->oend[o]; ++p) {
        const int c1 = m->pc0[p] + m->pw[p];
        GPK_TRY(predict_diag(m, p, c1, dR, dV, s));
        GPK_TRY(predict_update(m, p, c1, c1, cend - c1, dR, dV, s));
      }
      GPK_TRY(predict_outer_update(m, o, cend, cend, np_ - cend, dR, dV, s));
    }
    for (int o = 0; o < m->nO; ++o) {
      const int c0 = m->oc0[o], W = m->ow[o];
      GemmNT g{};
      g.A = dV + c0; g.lda = np_;
      g.B = dV + c0; g.ldb = np_;
      g.C = dNK; g.ldc = np_;
      g.m = c0 + W; g.n = c0 + W; g.k = W; g.sub = 1; g.lower = 1;
      GPK_TRY(launch_gemm_auto(g, s));
    }
    timer_end("grad_inv", s);
    timer_begin("grad_terms", s);
    GPK_HIP(hipMemcpyAsync(dalpha, m->dw, (size_t)np_ * 8, hipMemcpyDeviceToDevice, s));
    GPK_TRY(trsv_backward(m, dalpha));
    GPK_TRY(launch_nlml_grad_terms(m->dX, N, d, dNK, np_, dalpha, m->se, dpart, dout, s));
    timer_end("grad_terms", s);
    std::vector<double> t(d + 2);
    GPK_HIP(hipMemcpyAsync(t.data(), dout, (size_t)(d + 2) * 8, hipMemcpyDeviceToHost, s));
    GPK_HIP(hipStreamSynchronize(s));
    for (int k = 0; k < d; ++k) h_grad[k] = 0.5 * t[k];
    h_grad[d] = t[d];
    h_grad[d + 1] = m->sn2 * t[d + 1];
    return 0;
  };
  rc = body();
  if (rc) (void)hipStreamSynchronize(s);
  cleanup();
  return rc;
}

// ------------------------------------------------------------------ prediction entry shared by gpk_predict / bench
// M test points sharded over the P ranks of a block-cyclic model (every rank holds all of L): rank q solves the
// points [q Mc, (q+1) Mc), the [mean | var] slices are all-gathered.  ms[0..n) as in potrf_group; every local rank
// writes the gathered result to its own h_mean[i] / h_var[i].
static int group_predict_alloc(gpk_model* const* ms, int n, const double* dZ, int M, GroupPredict* gp) {
  if (n < 1 || !ms || !ms[0]) return gpk_set_err(-1, "predict_group: empty group");
  const int P = ms[0]->P;
  if (n > 1 && n != P) return gpk_set_err(-1, "predict_group: a group of %d needs the %d virtual ranks", n, P);
  gp->dZ = dZ;
  gp->M = M;
  gp->Mc = (M + P - 1) / P;  // test points per rank
  gp->Mpad = round_up(gp->Mc, 128);
  gp->bufs.assign(n, GroupPredict::Bufs());
  const size_t szR = (size_t)gp->Mpad * ms[0]->Npad, slice = (size_t)2 * gp->Mpad;
  int rc = 0;
  for (int i = 0; i < n && !rc; ++i) {
    rc |= dev_alloc_t(&gp->bufs[i].dR, szR);
    rc |= dev_alloc_t(&gp->bufs[i].dV, szR);
    rc |= dev_alloc_t(&ms[i]->shard_loc, slice);
    rc |= dev_alloc_t(&gp->bufs[i].dall, slice * P);
  }
  return rc;
}
static void group_predict_free(gpk_model* const* ms, int n, GroupPredict* gp) {
  const int P = ms[0]->P;
  const size_t szR = (size_t)gp->Mpad * ms[0]->Npad, slice = (size_t)2 * gp->Mpad;
  for (int i = 0; i < n && i < (int)gp->bufs.size(); ++i) {
    if (gp->bufs[i].dR) dev_free(gp->bufs[i].dR, szR * 8);
    if (gp->bufs[i].dV) dev_free(gp->bufs[i].dV, szR * 8);
    if (gp->bufs[i].dall) dev_free(gp->bufs[i].dall, slice * P * 8);
    if (ms[i]->shard_loc) dev_free(ms[i]->shard_loc, slice * 8);
    ms[i]->shard_loc = nullptr;
  }
}
// all-gather of the per-rank [mean | var] slices (m->shard_loc, filled on s_main) and read-back into every local rank's host arrays
static int group_predict_gather(gpk_model* const* ms, int n, GroupPredict* gp, double* const* h_mean, double* const* h_var) {
  hipStream_t s = G.s_main;
  const int P = ms[0]->P, M = gp->M, Mc = gp->Mc, Mpad = gp->Mpad;
  const size_t slice = (size_t)2 * Mpad;
  std::vector<std::vector<double>> hall(n, std::vector<double>(slice * P));
  for (int i = 0; i < n; ++i) {
    gpk_model* m = ms[i];
    if (P == 1) {
      GPK_HIP(hipMemcpyAsync(gp->bufs[i].dall, m->shard_loc, slice * 8, hipMemcpyDeviceToDevice, s));
    } else if (m->virtual_rank) {
      if (!m->vgroup) return gpk_set_err(-1, "internal: virtual rank %d without a group", m->r);
      for (int q = 0; q < P; ++q) {
        if (m->vgroup[q]->shard_loc)
          GPK_HIP(hipMemcpyAsync(gp->bufs[i].dall + slice * q, m->vgroup[q]->shard_loc, slice * 8, hipMemcpyDeviceToDevice, s));
        else  // (a single rank replayed alone, gpk_bench_dist_rank_alone: the others hold no shard)
          GPK_HIP(hipMemsetAsync(gp->bufs[i].dall + slice * q, 0, slice * 8, s));
      }
    } else {
      GPK_TRY(comm_allgather(m->shard_loc, gp->bufs[i].dall, slice, s));
    }
    GPK_HIP(hipMemcpyAsync(hall[i].data(), gp->bufs[i].dall, slice * P * 8, hipMemcpyDeviceToHost, s));
  }
  if (hipStreamSynchronize(s) != hipSuccess) return gpk_set_err(-2, "predict: %s", hipGetErrorString(hipGetLastError()));
  for (int i = 0; i < n; ++i)
    for (int q = 0; q < P; ++q) {
      const int q0 = q * Mc;
      const int qc = std::min(Mc, M - q0);
      for (int j = 0; j < qc; ++j) {
        h_mean[i][q0 + j] = hall[i][slice * q + j];
        h_var[i][q0 + j] = hall[i][slice * q + Mpad + j];
      }
    }
  return 0;
}

int predict_group(gpk_model* const* ms, int n, const double* dZ, int M, double* const* h_mean, double* const* h_var) {
  hipStream_t s = G.s_main;
  GroupPredict gp;
  int rc = group_predict_alloc(ms, n, dZ, M, &gp);
  auto body = [&]() -> int {
    const int P = ms[0]->P, d = ms[0]->d;
    for (int i = 0; i < n; ++i) {  // local solve of this rank's slice
      gpk_model* m = ms[i];
      const int m0 = (P > 1) ? m->r * gp.Mc : 0;
      const int cnt = std::max(0, std::min(gp.Mc, M - m0));
      GPK_HIP(hipMemsetAsync(m->shard_loc, 0, (size_t)2 * gp.Mpad * 8, s));
      if (cnt > 0)
        GPK_TRY(predict_device(m, dZ + (size_t)m0 * d, cnt, round_up(cnt, 128), gp.bufs[i].dR, gp.bufs[i].dV, m->shard_loc,
                               m->shard_loc + gp.Mpad));
    }
    return group_predict_gather(ms, n, &gp, h_mean, h_var);
  };
  if (!rc) rc = body();
  if (rc) (void)hipStreamSynchronize(s);
  if (n >= 1 && ms && ms[0]) group_predict_free(ms, n, &gp);
  return rc;
}

// fit + sharded prediction of a block-cyclic job in one go: the substitutions ride on the panel events (fit_group_impl), the slices
// are gathered afterwards.  rc > 0: not positive definite (as fit_group).
int fit_predict_group(gpk_model* const* ms, int n, const double* dZ, int M, double* const* h_mean, double* const* h_var) {
  hipStream_t s = G.s_main;
  GroupPredict gp;
  int rc = group_predict_alloc(ms, n, dZ, M, &gp);
  if (!rc) rc = fit_group_impl(ms, n, &gp);
  if (!rc) rc = group_predict_gather(ms, n, &gp, h_mean, h_var);
  if (rc) (void)hipStreamSynchronize(s);
  if (n >= 1 && ms && ms[0]) group_predict_free(ms, n, &gp);
  return rc;
}

int predict_sharded(gpk_model* m, const double* dZ, int M, double* h_mean, double* h_var) {
  if (m->virtual_rank && m->P > 1) return gpk_set_err(-1, "internal: virtual ranks predict as a group (predict_group)");
  gpk_model* one = m;
  return predict_group(&one, 1, dZ, M, &h_mean, &h_var);
}
