"""The flow of the reference's GP_preference_demo.py (seeded pairs :7-44, Laplace mode :60-67) on the MI355X core, without
the plotting, plus the two things its comments leave as pseudocode: the hyper-parameter search (`op.fmin(prefGP.calc_nlml,
theta0)`, commented out at :65) and the latent posterior over x (the demo only plots the mode at the training points)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropin"))
import GPpref  # noqa: E402  (dropin/GPpref: Python 3, no GPy, GPU core)


def run(verbose=True, optimise=False):
    log_hyp = np.log([0.1, 1.0, 0.1])   # length_scale, sigma_f, sigma_probit
    np.random.seed(1)
    n_train = 20
    true_sigma = 0.05
    delta_f = 1e-5

    def true_function(x):
        return np.sin(x * 2 * np.pi + np.pi / 4) + 0.2

    def obs_function(x, sigma):
        return true_function(x) + np.random.normal(scale=sigma, size=x.shape)

    def noisy_preference_rank(uv, sigma):
        fuv = obs_function(uv, sigma)
        y = -1 * np.ones((fuv.shape[0], 1), dtype='int')
        y[fuv[:, 1] > fuv[:, 0]] = 1
        return y, fuv

    x_plot = np.linspace(0.0, 1.0, 101, dtype='float')
    x_train = np.random.random((2 * n_train, 1))
    uvi_train = np.random.choice(range(2 * n_train), (n_train, 2), replace=False)
    uv_train = x_train[uvi_train][:, :, 0]
    y_train, fuv_train = noisy_preference_rank(uv_train, true_sigma)

    prefGP = GPpref.PreferenceGaussianProcess(x_train, uvi_train, y_train, delta_f=delta_f)
    theta = log_hyp
    if optimise:
        import scipy.optimize as op
        theta = op.fmin(prefGP.calc_nlml, log_hyp, disp=False, maxiter=60)
    f, lml = prefGP.calc_laplace(theta)
    mean, var = prefGP.predict_latent(x_plot.reshape(-1, 1))
    if verbose:
        print("lml %.10g after %d Newton iterations (jitter %g)" % (lml, prefGP.last_iterations, prefGP.last_eps))
        print("correlation of the latent mean with the true function on the grid: %.3f" % np.corrcoef(mean, true_function(x_plot))[0, 1])
    return dict(f=f, lml=lml, mean=mean, var=var, x_plot=x_plot, truth=true_function(x_plot), theta=theta, iters=prefGP.last_iterations)


if __name__ == "__main__":
    run()
