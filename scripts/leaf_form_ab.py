"""A/B of the leaf's two forms (slot-sized beside a loaded chip / full-tile once few rows remain: option leaf_fat_rows) and
of the polite GEMMs, one process per mode on one box.  (Round 3 used an earlier form of this script to compare the CU-masked
update stream of rounds 1-2 with the unmasked layout: gpurun logs quoted in DESIGN.md section 6.)
   python scripts/leaf_form_ab.py            -> runs every mode as a child process and prints one table
   python scripts/leaf_form_ab.py child ...  -> one mode (internal)"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SIZES = [(16384, 8, 1024, 10), (4096, 4, 1024, 40), (8192, 8, 1024, 20), (32768, 8, 1024, 3)]


def child(opts):
    from gptest_amd import _lib
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from gpk_hooks import hooks  # libgpk_hooks.so: probes and test hooks live outside the product library
    from bench import synthetic_problem

    _lib.init()
    l = _lib.lib()
    for o in opts:
        k, v = o.split("=")
        _lib.set_option(k, float(v))
    P = _lib.c_double_p
    us = ctypes.c_double()
    hooks().gpk_bench_leaf(0, 400, ctypes.byref(us))
    ua, uc = ctypes.c_double(), ctypes.c_double()
    hooks().gpk_bench_leaf_contended(200, ctypes.byref(ua), ctypes.byref(uc))
    print("   leaf alone %.1f us; on the chain stream %.1f us, beside back-to-back updates %.1f us" % (us.value, ua.value, uc.value), flush=True)
    hooks().gpk_bench_leaf_contended((8 << 16) | 100, ctypes.byref(ua), ctypes.byref(uc))  # phase stamps of the last leaf of each run
    for (N, d, M, reps) in SIZES:
        X, y, Z, lh = synthetic_problem(N, d, M)
        ms, nl = ctypes.c_double(), ctypes.c_double()
        mean, var = np.zeros(M), np.zeros(M)
        best = []
        for rep in range(3):
            _lib.set_option("timing", 1.0 if rep == 2 else 0.0)
            rc = l.gpk_bench_posterior(X.ctypes.data_as(P), y.ctypes.data_as(P), N, d, lh.ctypes.data_as(P), Z.ctypes.data_as(P), M, 1, reps,
                                       ctypes.byref(ms), ctypes.byref(nl), mean.ctypes.data_as(P), var.ctypes.data_as(P))
            assert rc == 0, _lib.last_error()
            best.append(ms.value / reps)
        a, c, w = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        l.gpk_last_timing(b"trailing", ctypes.byref(a), ctypes.byref(c))
        l.gpk_last_timing_work(b"trailing", ctypes.byref(w))
        tf = w.value / (a.value * 1e-3) / 1e12 if a.value > 0 else 0.0
        print("   N=%5d: %8.3f / %8.3f / %8.3f ms   trailing %.1f TFLOP/s (frac %.3f, %d launches)  nlml=%.12g" %
              (N, best[0], best[1], best[2], tf, tf / 78.6, int(c.value), nl.value), flush=True)
    _lib.shutdown()


MODES = [("full-tile leaf once <= %d rows remain, polite GEMMs" % r, {}, ["leaf_fat_rows=%d" % r]) for r in (-1, 4096, 6144, 8192, 1 << 30)] + [
    ("full-tile leaf once <= 6144 rows remain, GEMMs that never pause", {}, ["leaf_fat_rows=6144", "gemm_polite=0"])]

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2:])
        sys.exit(0)
    extra = sys.argv[1:]
    for rnd in range(1):
        for name, env, opts in MODES:
            print("[%d] %s  env=%s opts=%s" % (rnd, name, env, opts + extra), flush=True)
            e = dict(os.environ, **env)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"] + opts + extra, env=e, timeout=900)
            if p.returncode != 0:
                print("   child exited with %d" % p.returncode, flush=True)
