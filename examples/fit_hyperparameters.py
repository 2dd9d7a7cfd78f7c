"""Hyper-parameter fit with the analytic gradient: the demo's 1-D data (GP_regression_demo.py:6-20) fitted by
L-BFGS through ``compute_likelihood_gradient`` instead of Nelder-Mead through ``compute_likelihood``
(GP_regression_demo.py:29).  Both must land on the same optimum; the gradient needs an order of magnitude
fewer factorisations.  ``python examples/fit_hyperparameters.py [N d]`` runs a larger synthetic problem."""
import os
import sys
import time

import numpy as np
import scipy.optimize as op

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gptest_amd import GPr  # noqa: E402


def demo_data():
    np.random.seed(0)
    c = np.array([3, 5, -9, -3, 2], float)
    x = np.random.random(20)
    y = np.polyval(c, x) + np.random.normal(0, 0.1, len(x))
    return x, y, np.log([1, 1, 0.1])


def fit(x, y, log_hyp, verbose=True):
    gp = GPr.GaussianProcess(log_hyp, 0, 0, "SE", "zero", "zero", x, y)
    calls = [0]

    def fg(h):
        calls[0] += 1
        f, g = gp.compute_likelihood_gradient(h)
        return float(f.flat[0]), g

    t0 = time.perf_counter()
    res = op.minimize(fg, log_hyp, jac=True, method="L-BFGS-B")
    dt = time.perf_counter() - t0
    if verbose:
        print("L-BFGS: nlml %.6f after %d gradient evaluations (%.3f s), log_hyp %s" % (res.fun, calls[0], dt, res.x))
    return res.x, float(res.fun), calls[0]


if __name__ == "__main__":
    if len(sys.argv) >= 3:
        N, d = int(sys.argv[1]), int(sys.argv[2])
        rng = np.random.RandomState(5)
        x = rng.rand(N, d)
        y = np.sin(3 * x.sum(axis=1)) + 0.1 * rng.randn(N)
        fit(x, y, np.log(np.r_[np.ones(d), 1.0, 0.3]))
    else:
        x, y, lh = demo_data()
        hg, fg_, ng = fit(x, y, lh)
        gp = GPr.GaussianProcess(lh, 0, 0, "SE", "zero", "zero", x, y)
        hs, fs, _, ns, _ = op.fmin(gp.compute_likelihood, lh, full_output=True, disp=False)
        print("Nelder-Mead: nlml %.6f after %d evaluations, log_hyp %s" % (fs, ns, hs))
