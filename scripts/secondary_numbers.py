"""Secondary timings quoted in README.md / DESIGN.md: nlml gradient after a fit at N = 16384, gpk_solve with 512 right-hand
sides at N = 4096, the preference Laplace loop at 8192 pairs (inputs shaped like GP_preference_demo.py:39-44: nx = 2 * pairs,
pairs without replacement, labels from a noisy sine).  Usage: python scripts/secondary_numbers.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import GPpref, GPr, _lib  # noqa: E402
from bench import synthetic_problem  # noqa: E402

_lib.init()
l = _lib.lib()
P = _lib.c_double_p

X, y, Z, lh = synthetic_problem(16384, 8, 16)
h, nlml = GPr.fit_handle(X, y, lh)
g = np.zeros(10)
for rep in range(3):
    t0 = time.perf_counter()
    _lib.check(l.gpk_nlml_grad(h.ptr, g.ctypes.data_as(P)), "grad")
    t = time.perf_counter() - t0
print("nlml gradient, N = 16384, d = 8 (after the fit): %.1f ms" % (t * 1e3), flush=True)
del h

X, y, Z, lh = synthetic_problem(4096, 4, 16)
h, _ = GPr.fit_handle(X, y, lh)
B = np.ascontiguousarray(np.random.RandomState(0).randn(4096, 512))
out = np.empty_like(B)
for rep in range(3):
    t0 = time.perf_counter()
    _lib.check(l.gpk_solve(h.ptr, B.ctypes.data_as(P), 512, out.ctypes.data_as(P)), "solve")
    t = time.perf_counter() - t0
print("gpk_solve, nrhs = 512, N = 4096 (host buffers in and out): %.1f ms" % (t * 1e3), flush=True)
del h

rng = np.random.RandomState(1)
npairs, d = 8192, 4
x = rng.random_sample((2 * npairs, d))
uvi = rng.choice(2 * npairs, (npairs, 2), replace=False)
fx = np.sin(2.0 * np.pi * x.mean(axis=1)) + 0.05 * rng.randn(2 * npairs)
yp = np.where(fx[uvi[:, 1]] > fx[uvi[:, 0]], 1.0, -1.0).reshape(-1, 1)
lhp = np.log([0.3] * d + [1.0, 0.1])
gp = GPpref.PreferenceGaussianProcess(x, uvi, yp, delta_f=1e-6)
for rep in range(2):
    t0 = time.perf_counter()
    f, lml = gp.calc_laplace(lhp)
    t = time.perf_counter() - t0
print("preference Laplace loop, 8192 pairs (nx = 16384), d = 4: %.1f ms, %d Newton iterations, converged %s, lml %.6f"
      % (t * 1e3, gp.last_iterations, gp.last_converged, lml), flush=True)
# accounting (BASELINE config 5): one Cholesky of K + eps I at nx (nx^3 / 3 flops) and, per Newton iteration, one of
# B = I + C^1/2 (D K D^T) C^1/2 at npairs, two GEMVs with the kept copy of K (8 nx^2 bytes each) and the build of B (16 npairs^2 bytes)
nx = 2 * npairs
flop = nx ** 3 / 3.0 + gp.last_iterations * npairs ** 3 / 3.0
byts = gp.last_iterations * (2 * 8.0 * nx * nx + 16.0 * npairs * npairs)
print("   accounting: %.2f Tflop of fp64 MFMA work (K: %.2f, %d x B: %.2f each) -> %.1f TFLOP/s = %.2f of 78.6 over the loop; %.1f GB of GEMV / build-B "
      "traffic = %.2f ms at 8 TB/s.  Both factorisations are chain-bound sizes (Npad = 16384 once, 8192 per iteration: 7.9-8.3 ms each alone)"
      % (flop / 1e12, nx ** 3 / 3.0 / 1e12, gp.last_iterations, npairs ** 3 / 3.0 / 1e12, flop / t / 1e12, flop / t / 78.6e12, byts / 1e9, byts / 8e12 * 1e3), flush=True)
_lib.shutdown()
