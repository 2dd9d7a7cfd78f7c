"""Robustness run: many fits of random sizes in one process, failures interleaved, pooled memory watched."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gptest_amd import GPr, GPpref, _lib
from oracle import gpr_oracle as orc
_lib.init()
rng = np.random.RandomState(7)
worst = 0.0
t0 = time.time()
nfail = 0
for it in range(150):
    N = int(rng.choice([1, 2, 5, 17, 127, 128, 129, 300, 511, 640, 1000, 1500, 2049, 3000]))
    d = int(rng.randint(1, 9))
    x, y, z, lh = orc.synthetic_problem(N, d, M=int(rng.randint(1, 300)), seed=it)
    if it % 10 == 9:
        # not positive definite: duplicated points and (numerically) zero noise
        xb = np.vstack([x, x])[: max(2, N)]
        yb = np.concatenate([y, y])[: max(2, N)]
        lhb = lh.copy(); lhb[-1] = -40.0
        try:
            GPr.GaussianProcess(lhb, 0, 0, "SE", "zero", "zero", xb, yb).compute_likelihood(lhb)
        except np.linalg.LinAlgError:
            nfail += 1
        continue
    gp = GPr.GaussianProcess(lh, 0, 0, "SE", "zero", "zero", x, y)
    mean, var = gp.compute_prediction(z)
    nl = gp.compute_likelihood(lh).flat[0]
    if it % 7 == 0 and N >= 3:
        gp.compute_likelihood_gradient(lh)
    if N <= 1000:
        rn, rm, rv = orc.posterior_efficient(lh, x, y, z)
        e = max(abs(nl - rn) / max(1.0, abs(rn)), np.abs(mean - rm).max() / max(1e-300, np.abs(rm).max()), np.abs(var - rv).max() / max(1e-300, np.abs(rv).max()))
        worst = max(worst, e)
print("150 iterations in %.1f s: worst relative error %.2e, %d expected LinAlgErrors, pooled HBM %.1f MiB" % (
    time.time() - t0, worst, nfail, _lib.get_option("pool_bytes") / 2**20))
assert worst < 1e-6
_lib.shutdown(watchdog_s=10.0)
