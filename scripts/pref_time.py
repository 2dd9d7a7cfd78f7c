import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import GPpref, _lib
_lib.init()
for o in sys.argv[1:]:
    k, v = o.split("="); _lib.set_option(k, float(v))
rng = np.random.RandomState(1)
npairs, d = 8192, 4
x = rng.random_sample((2 * npairs, d))
uvi = rng.choice(2 * npairs, (npairs, 2), replace=False)
fx = np.sin(2.0 * np.pi * x.mean(axis=1)) + 0.05 * rng.randn(2 * npairs)
yp = np.where(fx[uvi[:, 1]] > fx[uvi[:, 0]], 1.0, -1.0).reshape(-1, 1)
lhp = np.log([0.3] * d + [1.0, 0.1])
gp = GPpref.PreferenceGaussianProcess(x, uvi, yp, delta_f=1e-6)
for rep in range(3):
    t0 = time.perf_counter(); f, lml = gp.calc_laplace(lhp); t = time.perf_counter() - t0
    print(sys.argv[1:], "%.1f ms" % (t * 1e3), flush=True)
