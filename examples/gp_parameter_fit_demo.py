"""The flow of the reference's GP_parameter_fit.py (:8-33 cost field, 200 noisy samples, RBF kernel, hyper-parameter
optimisation; :50-63 a model that receives one more observation per video frame) on the MI355X core, without the
plotting.  `gpy_compat` serves the GPy calls; from the second frame on every `set_XY` extends the Cholesky factor held in
HBM (`gpk_append`) instead of refactoring."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropin"))
import GPy  # noqa: E402  (dropin/GPy: the calls of GP_parameter_fit.py on the GPU core)

mean_value = 3


def explore_cost_function(a, b):          # GP_parameter_fit.py:9-20
    cost = mean_value
    cost += 10 * math.exp(-math.sqrt((a - 40) ** 2 + (b - 40) ** 2) / 16)
    cost += 7 * math.exp(-math.sqrt((a - 10) ** 2 + (b - 90) ** 2) / 12)
    cost += 4 * math.exp(-math.sqrt((a - 80) ** 2 + (b - 60) ** 2) / 32)
    cost += 7 * math.exp(-math.sqrt((a + 20) ** 2 + (b - 50) ** 2) / 32)
    cost += 7 * math.exp(-math.sqrt((a - 120) ** 2 + (b - 50) ** 2) / 32)
    cost += 12 * math.exp(-math.sqrt((a - 80) ** 2 + (b - 20) ** 2) / 8)
    cost += 5 * math.exp(-math.sqrt((a - 60) ** 2 + (b - 80) ** 2) / 10)
    cost += 3 * math.exp(-math.sqrt((a - 90) ** 2 + (b - 90) ** 2) / 20)
    return cost


def run(verbose=True, n_points=200, frames=40, seed=0):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0., 100., (n_points, 2))
    Y = np.array([[explore_cost_function(a, b)] for a, b in X]) + rng.randn(n_points, 1) * 0.25
    kernel = GPy.kern.RBF(input_dim=2, variance=10., lengthscale=20.)
    gpm = GPy.models.GPRegression(X, Y - mean_value, kernel)
    gpm.optimize(messages=verbose)
    Xtemp, Ytemp = np.meshgrid(np.arange(100), np.arange(100))
    Xfull = np.vstack([Xtemp.ravel(), Ytemp.ravel()]).transpose().astype(float)
    truth = np.array([explore_cost_function(a, b) for a, b in Xfull])
    errs = []
    for ii in range(frames):              # GP_parameter_fit.py:61-63
        gpm.set_XY(X[0:ii + 1, :], Y[0:ii + 1] - mean_value)
        Yfull, varYfull = gpm.predict(Xfull)
        errs.append(float(np.sqrt(np.mean((Yfull.ravel() + mean_value - truth) ** 2))))
    if verbose:
        print(gpm)
        print("field RMSE after 1 / %d observations: %.3f / %.3f; appended %d points, %d refits" % (
            frames, errs[0], errs[-1], gpm.appended_points, gpm.refits))
    return dict(rmse=errs, appended=gpm.appended_points, refits=gpm.refits, model=gpm)


if __name__ == "__main__":
    run()
