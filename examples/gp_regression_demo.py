"""The reference's 1-D regression demo flow (GP_regression_demo.py:6-48) on the MI355X path: seeded data,
Nelder-Mead over the three log hyper-parameters through ``compute_likelihood`` (331 evaluations), one
``compute_prediction``.  No plotting.  With ``dropin/`` on PYTHONPATH the reference's own script runs unchanged
instead; this file exists because the reference cannot travel to the GPU box."""
import os
import sys

import numpy as np
import scipy.optimize as op

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dropin"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import GPr  # noqa: E402  (resolves to dropin/GPr.py -> gptest_amd.GPr)


def run(verbose=True):
    np.random.seed(0)
    c = np.array([3, 5, -9, -3, 2], float)
    x_train = np.random.random(20)
    y_train = np.polyval(c, x_train) + np.random.normal(0, 0.1, len(x_train))
    x_test = np.arange(0, 1, 0.01, float)
    log_hyp = np.log([1, 1, 0.1])
    initGP = GPr.GaussianProcess(log_hyp, 0, 0, "SE", "zero", "zero", x_train, y_train)
    opt_log_hyp, fopt, iters, funcalls, _ = op.fmin(initGP.compute_likelihood, log_hyp, full_output=True, disp=verbose)
    optGP = GPr.GaussianProcess(opt_log_hyp, 0, 0, "SE", "zero", "zero", x_train, y_train)
    y_test, cov_y = optGP.compute_prediction(x_test)
    band = np.sqrt(cov_y)  # must be real, as the demo plots y +- sqrt(cov) (GP_regression_demo.py:52-53)
    return dict(opt_log_hyp=opt_log_hyp, nlml=float(fopt), iters=iters, funcalls=funcalls, y_test=y_test, cov_y=cov_y,
                band=band)


if __name__ == "__main__":
    r = run()
    print("opt_log_hyp", r["opt_log_hyp"], "nlml", r["nlml"], "evaluations", r["funcalls"])
    print("y_test[:3]", r["y_test"][:3], "cov_y[:3]", r["cov_y"][:3])
