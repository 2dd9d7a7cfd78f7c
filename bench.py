#!/usr/bin/env python3
"""Headline benchmark: GP posterior solves / second at N=16384, d=8 (BASELINE.json).

One "step" = one GP posterior = compute_likelihood (GPr.py:57-69) + compute_prediction for
M=1024 test points (GPr.py:45-54) sharing one K build and one Cholesky, on the SURVEY 8(d)
synthetic inputs, with X, y, Z already resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: the SAME N=16384 posterior is sharded (block-cyclic panels, RCCL panel broadcast over xGMI):
total work fixed -> "scaling": "strong".  The launcher only provides RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*;
the ncclUniqueId travels over a TCP socket, barriers and the max over ranks go through RCCL.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6  # vendor dense fp64 matrix peak, MI355X (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
PEAK_HBM_GBS = 8000.0


def synthetic_problem(N, d, M, seed=1234):
    """SURVEY.md 8(d) inputs (same generator as oracle.gpr_oracle.synthetic_problem)."""
    rng = np.random.RandomState(seed)
    X = rng.random_sample((N, d))
    w = rng.randn(d)
    y = np.sin(X @ w) + 0.1 * rng.randn(N)
    Z = rng.random_sample((M, d))
    log_hyp = np.log([0.7] * d + [1.0, 0.1])
    return X, y, Z, log_hyp


def _blas_info():
    cores, name = None, None
    try:
        from threadpoolctl import threadpool_info

        blas = [p for p in threadpool_info() if p.get("user_api") == "blas"]
        if blas:
            cores = max(p.get("num_threads", 1) for p in blas)
            name = ", ".join(sorted({"%s %s" % (p.get("internal_api", "?"), p.get("version", "")) for p in blas}))
    except Exception:
        pass
    if not cores:
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            cores = os.cpu_count() or 1
    return cores, name


def cpu_baseline(N, d, M, budget_n=4096):
    """The oracle timed on the host cores (rank 0, N = 1 only):
    * the verbatim-cost restatement of GPr.py (inv + cholesky + two LU solves against eye(n), GPr.py:45-69) on a bounded
      sample N_s = budget_n, scaled by (N/N_s)^3 (every dominant term is n^3; the full size would take ~14 min) -- the
      reference's own algorithm, the headline "port" figure;
    * the efficient flavour (SciPy cho/triangular solves, one shared factorisation: the algorithm the HIP path
      implements) MEASURED AT THE METRIC'S SIZE, one posterior at N (about a minute on the box's 64 threads)."""
    import resource

    from oracle import gpr_oracle as orc

    cores, blas = _blas_info()
    X, y, Z, lh = orc.synthetic_problem(min(budget_n, N), d, M)
    t0 = time.perf_counter()
    orc.posterior_verbatim(lh, X, y, Z)
    t = time.perf_counter() - t0
    scale = (N / float(len(X))) ** 3
    X2, y2, Z2, lh2 = orc.synthetic_problem(N, d, M)
    t0 = time.perf_counter()
    orc.posterior_efficient(lh2, X2, y2, Z2)
    t2 = time.perf_counter() - t0
    rss_gib = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2.0 ** 20
    return {
        "value": 1.0 / (t * scale),
        "unit": "posteriors/s",
        "cores": cores,
        "blas": blas,
        "kind": "port",
        "sample": "one verbatim-cost posterior (inv + cholesky + 2 LU solves, GPr.py:45-69) at N=%d d=%d M=%d "
                  "took %.2f s; n^3-scaled x%.0f to N=%d" % (len(X), d, M, t, scale, N),
        "efficient": {
            "value": 1.0 / t2,
            "unit": "posteriors/s",
            "cores": cores,
            "kind": "port",
            "sample": "measured at N=%d: one efficient posterior (scipy cholesky + triangular solves, one shared factorisation) "
                      "at N=%d d=%d M=%d took %.2f s, not scaled; peak RSS of the process %.1f GiB" % (N, N, d, M, t2, rss_gib),
        },
    }


def parity_check(N, d, M, nlml, mean, var, lh):
    """Compare this run's outputs with the committed full-size fixture (tests/golden/full_*.npz, written by
    tests/golden/make_fullsize.py from the CPU oracle).  Returns (checked, detail, ok); ok is False on a mismatch
    beyond the north-star bar of 1e-6 relative: a fast kernel with wrong results must not print a headline
    (main() then exits with status 4 after the closing barrier, so that no other rank is left waiting)."""
    name = {(16384, 8, 1024): "c3_16384_d8", (4096, 4, 1024): "c2_4096_d4", (32768, 8, 1024): "c4_32768_d8"}.get((N, d, M))
    if name is None:
        return False, "no committed fixture for N=%d d=%d M=%d" % (N, d, M), True
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_%s.npz" % name))
    ref = float(g["nlml"])
    e_nl = abs(nlml - ref) / abs(ref)
    e_mean = float(np.abs(mean - g["fz"]).max() / np.abs(g["fz"]).max())
    e_var = float(np.abs(var - g["cov_fz"]).max() / np.exp(lh[-2]) ** 2)
    detail = {"fixture": "tests/golden/full_%s.npz" % name, "rel_err_nlml": e_nl, "rel_err_mean": e_mean,
              "err_var_over_sf2": e_var, "tol": 1e-6}
    ok = bool(e_nl <= 1e-6 and e_mean <= 1e-6 and e_var <= 1e-6)
    return ok, detail, ok


def phase_split(l, _lib, X, y, Z, lh, N, d, M):
    """roofline.phases: one extra posterior AFTER the timed region with the library's device-side stamps on (a one-wave kernel
    behind every panel of the chain; no profiler), split where the panel chain leaves the widest (512-column) panels: up to
    there the bulk updates are the bound, behind it the chain of dependent leaf / panel kernels is.  Work per phase is what the
    schedule issues in it: trailing updates m^2 w and panel solves of its panels, the prediction's substitution updates (the far
    part of the early panels counted where the deferred schedule runs it), the lazy-corner catch-up where it runs."""
    P = _lib.c_double_p
    _lib.check(l.gpk_set_option(b"stamps", 1.0), "stamps")
    ms, nl = ctypes.c_double(), ctypes.c_double()
    mean, var = np.zeros(M), np.zeros(M)
    try:
        rc = l.gpk_bench_posterior(X.ctypes.data_as(P), y.ctypes.data_as(P), N, d, lh.ctypes.data_as(P), Z.ctypes.data_as(P), M, 1, 1,
                                   ctypes.byref(ms), ctypes.byref(nl), mean.ctypes.data_as(P), var.ctypes.data_as(P))
        if rc != 0:
            return None
        g = lambda k: l.gpk_get_option(k.encode())
        nP = int(g("last_panels"))
        c0 = [int(g("last_panel_c0_%d" % p)) for p in range(nP)]
        w = [int(g("last_panel_w_%d" % p)) for p in range(nP)]
        st = [g("stamp_panel_%d" % p) for p in range(nP)]
        t_start = g("stamp_start")
        if nP < 3 or any(x != x for x in st) or t_start != t_start:
            return None
        npad = c0[-1] + w[-1]
        mpad = (M + 127) // 128 * 128
        wmax = max(w)
        p1 = next((p for p in range(nP) if w[p] != wmax), nP)  # first panel that is not of the widest kind
        if p1 in (0, nP):
            return None
        pdc = int(g("pred_defer_col"))
        cb = int(g("potrf_corner_from"))
        work = [0.0, 0.0]
        for p in range(nP):
            ph = 0 if p < p1 else 1
            ce = c0[p] + w[p]
            mm = float(npad - ce)
            # factorisation: trailing update (the part behind a lazy-corner boundary runs with the catch-up launches: phase 2) + panel solves
            if cb and ce < cb:
                cm = float(npad - cb)
                work[ph] += (mm * mm - cm * cm) * w[p]
                work[1] += cm * cm * w[p]
            else:
                work[ph] += mm * mm * w[p]
            work[ph] += float(npad - c0[p]) * w[p] * w[p]
            # prediction: diagonal solve + update of the columns behind the panel (far part deferred to the boundary: phase of that panel)
            work[ph] += 2.0 * mpad * w[p] * w[p] / 2
            if pdc and ce <= pdc:
                work[ph] += 2.0 * mpad * (pdc - ce) * w[p]
                bph = 0 if pdc < c0[p1] else 1
                work[bph] += 2.0 * mpad * (npad - pdc) * w[p]
            else:
                work[ph] += 2.0 * mpad * mm * w[p]
        t1 = st[p1] - t_start          # the chain has left the widest panels
        t_all = ms.value               # one posterior, stamps on
        t2 = t_all - t1
        return {
            "bulk_bound": {"ms": t1, "tflop": work[0] / 1e12, "tflops": work[0] / (t1 * 1e-3) / 1e12, "panels": "%d x %d columns" % (p1, wmax)},
            "chain_bound": {"ms": t2, "tflop": work[1] / 1e12, "tflops": work[1] / (t2 * 1e-3) / 1e12,
                            "panels": "%d (narrower panels, tile pipeline)" % (nP - p1), "chain_done_ms": st[-1] - t_start},
            "whole_posterior": {"ms": t_all, "tflop": (work[0] + work[1]) / 1e12, "tflops": (work[0] + work[1]) / (t_all * 1e-3) / 1e12},
            "note": "one posterior after the timed region with device-side stamps on (no profiler); work = what the schedule issues in each phase",
        }
    finally:
        l.gpk_set_option(b"stamps", 0.0)


def secondary_configs(l, _lib):
    """BASELINE.json's other one-GPU configurations, measured after the headline's timed region so that the driver's record
    carries them too (about 1.5 s): configs[1] (GPr N=4096 d=4, parity against its committed fixture) and configs[4] (the
    preference Laplace loop at 8192 pairs, d=4; inputs shaped like GP_preference_demo.py:39-44, as scripts/secondary_numbers.py).
    Event timers off: these are plain wall-clock / device-event figures of the default schedule."""
    from gptest_amd import GPpref

    P = _lib.c_double_p
    _lib.check(l.gpk_set_option(b"timing", 0.0), "timing")
    out = {}
    N, d, M = 4096, 4, 1024
    X, y, Z, lh = synthetic_problem(N, d, M)
    ms, nl = ctypes.c_double(), ctypes.c_double()
    mean, var = np.zeros(M), np.zeros(M)
    steps = 40
    _lib.check(l.gpk_bench_posterior(X.ctypes.data_as(P), y.ctypes.data_as(P), N, d, lh.ctypes.data_as(P), Z.ctypes.data_as(P), M, 5, steps,
                                     ctypes.byref(ms), ctypes.byref(nl), mean.ctypes.data_as(P), var.ctypes.data_as(P)), "configs[1]")
    checked, detail, ok = parity_check(N, d, M, nl.value, mean, var, lh)
    flop = N ** 3 / 3.0 + float(M) * N * N
    out["configs_1"] = {"workload": "GPr regression N=4096 d=4 RBF fp64, M=1024 (BASELINE configs[1])", "value": steps / (ms.value * 1e-3),
                        "unit": "posteriors/s", "ms_per_step": ms.value / steps, "steps": steps, "parity_checked": checked, "parity": detail,
                        "tflops_whole_posterior": flop / (ms.value / steps * 1e-3) / 1e12,
                        "note": "latency-bound size: 32 dependent leaf steps of ~85 us (DESIGN.md section 6); N^3/3 + M N^2 flops per posterior"}
    if not ok:
        raise RuntimeError("configs[1] parity failure: %r" % (detail,))
    rng = np.random.RandomState(1)
    npairs, d = 8192, 4
    x = rng.random_sample((2 * npairs, d))
    uvi = rng.choice(2 * npairs, (npairs, 2), replace=False)
    fx = np.sin(2.0 * np.pi * x.mean(axis=1)) + 0.05 * rng.randn(2 * npairs)
    yp = np.where(fx[uvi[:, 1]] > fx[uvi[:, 0]], 1.0, -1.0).reshape(-1, 1)
    lhp = np.log([0.3] * d + [1.0, 0.1])
    gp = GPpref.PreferenceGaussianProcess(x, uvi, yp, delta_f=1e-6)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        f, lml = gp.calc_laplace(lhp)
        t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
    nx, it = 2 * npairs, gp.last_iterations
    flop = nx ** 3 / 3.0 + it * npairs ** 3 / 3.0
    byts = it * (2 * 8.0 * nx * nx + 16.0 * npairs * npairs)
    out["configs_4"] = {"workload": "GPpref preference Laplace loop, 8192 pairs (nx = 16384), d=4 (BASELINE configs[4]); calc_laplace through the Python surface, host buffers in and out",
                        "ms_per_loop": best * 1e3, "value": 1.0 / best, "unit": "Laplace fits/s", "newton_iterations": it, "converged": bool(gp.last_converged),
                        "lml": float(lml),
                        "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_FP64_MFMA_TFLOPS, "achieved": flop / best / 1e12, "frac": flop / best / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                                     "flop": flop, "hbm_bytes_gemv_and_build_B": byts, "hbm_ms_at_peak": byts / (PEAK_HBM_GBS * 1e9) * 1e3,
                                     "note": "one Cholesky of K + eps I at nx (nx^3/3) and per Newton iteration one of B = I + C^1/2 (D K D^T) C^1/2 at npairs (npairs^3/3), "
                                             "two GEMVs with the kept K (8 nx^2 B each) and the build of B (16 npairs^2 B); both factorisations are chain-bound sizes"},
                        "parity": "unpinned (the reference's GPpref.py is Python 2 + GPy and holds no fixture; tests compare with oracle/gppref_oracle.py)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--train-n", dest="n", type=int, default=16384)
    ap.add_argument("--dim", dest="d", type=int, default=8)
    ap.add_argument("--test-m", dest="m", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-probe", action="store_true",
                    help="skip roofline.sustained (its 24 extra update launches would be counted by the kernel-trace / PMC scripts)")
    ap.add_argument("--no-phases", action="store_true", help="skip roofline.phases (one extra posterior with device stamps after the timed region)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` block (BASELINE configs[1] and configs[4], ~1.5 s after the timed region)")
    ap.add_argument("--opt", action="append", default=[], help="libgpk option key=value (experiments; defaults are the product)")
    args = ap.parse_args()

    from gptest_amd import _lib, dist

    rank, local, world = dist.env_rank()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    # One process per GPU (torch.distributed.run only provides RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).
    # The ncclUniqueId travels over a plain TCP socket, barriers and the max-over-ranks go through RCCL:
    # torch is never imported (importing it first would also shadow ROCm's librccl/libamdhip64 with the
    # copies bundled in the torch wheel).
    dist.init_comm(rank, world, device=local)
    l = _lib.lib()
    N, d, M = args.n, args.d, args.m
    X, y, Z, lh = synthetic_problem(N, d, M)
    P = _lib.c_double_p
    ms = ctypes.c_double()
    nlml = ctypes.c_double()
    mean = np.zeros(M)
    var = np.zeros(M)
    _lib.check(l.gpk_set_option(b"timing", 1.0), "timing")
    for kv in args.opt:
        k, v = kv.split("=")
        _lib.check(l.gpk_set_option(k.encode(), float(v)), "option " + k)

    # barrier + device sync on both sides of the timed region (gpk_bench_posterior stages the inputs,
    # runs `warmup` untimed steps, then times exactly `steps` steps with HIP events on its stream).
    _lib.check(l.gpk_comm_barrier(), "barrier")
    _lib.check(l.gpk_sync(), "sync")
    t0 = time.perf_counter()
    rc = l.gpk_bench_posterior(X.ctypes.data_as(P), y.ctypes.data_as(P), N, d, lh.ctypes.data_as(P),
                               Z.ctypes.data_as(P), M, args.warmup, args.steps, ctypes.byref(ms),
                               ctypes.byref(nlml), mean.ctypes.data_as(P), var.ctypes.data_as(P))
    _lib.check(rc, "gpk_bench_posterior")
    _lib.check(l.gpk_sync(), "sync")
    wall = time.perf_counter() - t0
    t_ms = ms.value
    _lib.check(l.gpk_comm_barrier(), "barrier")
    if world > 1:
        tt = (ctypes.c_double * 1)(t_ms)
        _lib.check(l.gpk_comm_allreduce_max(tt, 1), "allreduce_max")
        t_ms = float(tt[0])

    def stage(name):
        a, c = ctypes.c_double(), ctypes.c_double()
        l.gpk_last_timing(name.encode(), ctypes.byref(a), ctypes.byref(c))
        return a.value, c.value

    def stage_work(name):
        a = ctypes.c_double()
        l.gpk_last_timing_work(name.encode(), ctypes.byref(a))
        return a.value

    parity_ok = True
    if rank == 0:
        checked, parity_detail, parity_ok = parity_check(N, d, M, nlml.value, mean, var, lh)
        nb = int(l.gpk_get_option(b"nb"))
        tr_ms, tr_n = stage("trailing")
        kb_ms, kb_n = stage("kbuild")
        # algorithmic flops of the timed trailing-update (SYRK) launches, m^2 * w each, accumulated by
        # libgpk next to the HIP-event durations of exactly those launches (look-ahead: the first block
        # column of every update runs on the panel stream and is not part of the timed launches).
        tr_flops = stage_work("trailing")
        kb_bytes = stage_work("kbuild")
        ach = tr_flops / (tr_ms * 1e-3) / 1e12 if tr_ms > 0 else 0.0
        # HBM traffic of that kernel per launch: PMC counters cannot be read from inside the process, so the
        # figure comes from the committed rocprofv3 --pmc passes of this same command (scripts/profile_round.sh)
        traffic = None
        traffic_note = None
        try:
            import glob

            newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_gemm_nt8.json")))[-1]
            with open(newest) as fh:
                tj = json.load(fh)
            if world == 1 and N == 16384:
                traffic = tj["hbm_bytes_per_launch"]
                traffic_note = ("%s: PMC FETCH_SIZE(x2, gfx950 correction)+WRITE_SIZE from two separate rocprofv3 --pmc passes of this "
                                "command, avg over the %d trailing-update launches of one factorisation; algorithmic %.3g B/launch"
                                % (os.path.basename(newest), tj["launches"], tj["algorithmic_bytes_per_launch"]))
        except Exception:
            pass
        mfma_pmc = None
        try:
            import glob

            with open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mfma_util_gemm_nt8.json")))[-1]) as fh:
                mj = json.load(fh)
            if world == 1 and N == 16384:
                mfma_pmc = {"mfma_util_percent": mj["mfma_util_percent"], "shader_clock_ghz": mj["shader_clock_ghz"],
                            "tflops": mj["tflops_in_this_pass"],
                            "note": "committed rocprofv3 --pmc pass of this command (SQ_VALU_MFMA_BUSY_CYCLES, "
                                    "SQ_INSTS_VALU_MFMA_MOPS_F64, GRBM_GUI_ACTIVE); counter collection serialises dispatches, so "
                                    "these launches ran ALONE there, not next to the panel chain and the prediction"}
        except Exception:
            pass
        flops = tr_flops / max(args.steps, 1)
        # the chip's fp64 MFMA pipe re-measured on THIS box after the timed region: an issue loop of
        # v_mfma_f64_16x16x4_f64 with exactly 4 waves pinned on every SIMD (gpk_bench_mfma_pinned), about 10 ms
        peak_measured = None
        if world == 1:
            try:
                pm = (ctypes.c_double * 6)()
                if l.gpk_bench_mfma_pinned(4, 60000, pm) == 0:
                    peak_measured = {"tflops": pm[0], "shader_clock_ghz": pm[1], "frac_of_it": (ach / pm[0]) if pm[0] else None,
                                     "note": "v_mfma_f64_16x16x4_f64 issue loop, 4 waves pinned per SIMD on all 256 CUs, run after the timed "
                                             "region on the update stream (no stream is CU-masked any more)"}
            except Exception:
                pass
        # ... and the shader clock the chip actually runs the update kernel at: a one-wave probe (s_memtime / s_memrealtime) on the
        # chain stream while the update stream runs back-to-back lower updates at K = 512 (~60 ms).  Under sustained fp64 MFMA load
        # the clock drops from the ~2.36 GHz of the memory-free issue loop above to ~2.0-2.2 GHz: the ceiling of a sustained update
        # is 256 CUs x 4 SIMDs x 32 flop x THAT clock, not the nominal 78.6 TFLOP/s `frac` is quoted against.
        sustained = None
        if world == 1 and peak_measured is not None and not args.no_clock_probe:
            try:
                ns = 8
                ghz = (ctypes.c_double * (ns + 1))()
                tfu = ctypes.c_double()
                if l.gpk_bench_clock_under_load(16128, 512, 24, ns, ghz, ctypes.byref(tfu)) == 0:
                    g = [x for x in list(ghz)[1:] if x > 0]
                    if g:
                        ghz_mean = sum(g) / len(g)
                        sustained = {"shader_clock_ghz_under_update": ghz_mean, "min": min(g), "idle": ghz[0],
                                     "ceiling_tflops_at_that_clock": 256 * 4 * 32 * ghz_mean * 1e-3,
                                     "update_alone_tflops": tfu.value,
                                     "frac_of_that_ceiling_in_situ": ach / (256 * 4 * 32 * ghz_mean * 1e-3),
                                     "note": "one-wave clock probe beside 24 back-to-back lower updates (m = 16128, K = 512), after the timed region"}
            except Exception:
                pass
        phases = None
        if world == 1 and not args.no_phases:
            try:
                phases = phase_split(l, _lib, X, y, Z, lh, N, d, M)
            except Exception:
                phases = None
        out = {
            "metric": "GP posterior solves/sec (N=%d, d=%d, M=%d; fit + predict sharing one Cholesky)" % (N, d, M),
            "value": args.steps / (t_ms * 1e-3),
            "unit": "posteriors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": t_ms / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "GPr regression N=%d d=%d RBF fp64, M=%d test points%s" % (
                N, d, M, " (BASELINE configs[2], the size the metric is quoted on)" if (N, d) == (16384, 8) else
                " (BASELINE configs[1])" if (N, d) == (4096, 4) else ""),
                       "panel_schedule": "512/256/128 columns by remaining size (nb_auto), tile pipeline for the last 4096 columns" if world == 1 else "512/256/128 columns by remaining size (nb_auto), ownership blocks of 512 columns",
                       "parallelism": "1 GPU" if world == 1 else "block-cyclic panels x%d, RCCL broadcast" % world},
            "roofline": {"kernel": "gemm_nt8_kernel (Cholesky trailing update, fp64 MFMA 16x16x4)", "bound": "mfma",
                         "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_FP64_MFMA_TFLOPS, "peak_measured": peak_measured, "sustained": sustained,
                         "traffic": traffic, "traffic_note": traffic_note,
                         "mfma_pmc_serialised": mfma_pmc,
                         "phases": phases,
                         "launches": tr_n, "avg_launch_ms": tr_ms / tr_n if tr_n else None,
                         "alg_flops_per_step_in_timed_launches": flops},
            "kbuild": {"kernel": "se_scale_rows_kernel + se_build_srow_kernel (scalar-operand form, timed together)", "bound": "hbm", "ms": kb_ms / kb_n if kb_n else None, "unit": "GB/s",
                       "peak": PEAK_HBM_GBS, "achieved": kb_bytes / (kb_ms * 1e-3) / 1e9 if kb_ms else None,
                       "frac": kb_bytes / (kb_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if kb_ms else None,
                       "note": "only the lower-triangle tiles are written: 4*N^2 + 8*N*d algorithmic bytes per build"},
            "nlml": nlml.value,
            "parity_checked": checked,
            "parity": parity_detail,
            "wall_s_incl_staging": wall,
        }
        if world == 1 and not args.no_secondary and (N, d, M) == (16384, 8, 1024):
            try:
                out["secondary"] = secondary_configs(l, _lib)
            except Exception as e:  # never lose the headline over a secondary figure
                out["secondary"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only: the other ranks would idle in the barrier
            out["cpu_baseline"] = cpu_baseline(N, d, M)
        if parity_ok:
            print(json.dumps(out), flush=True)
        else:
            sys.stderr.write("bench.py: PARITY FAILURE, no result line printed: %s\n" % json.dumps(out))
            sys.stderr.flush()
    _lib.check(l.gpk_comm_barrier(), "barrier")
    l.gpk_comm_destroy()
    # everything is measured and printed: release the GPU objects explicitly (a profiler attached to this process
    # needs that for a clean exit), but never let the release keep the process alive
    _lib.shutdown()
    if rank == 0 and not parity_ok:
        sys.exit(4)


if __name__ == "__main__":
    main()
