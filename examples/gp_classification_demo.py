"""Simple 1D GP classification example -- the flow GP_classification_demo.py:5-50 sets up (seeded data, class
probabilities Phi(true_function), 20 noisy labels) carried through to the part the reference never wrote: the
Laplace mode, its approximate log marginal likelihood over a sweep of length scales and the predictive class
probabilities, all on the GPU core (gptest_amd.GPc)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import GPc  # noqa: E402


def run(verbose=True):
    from scipy.special import ndtr as std_norm_cdf

    np.random.seed(0)

    def true_function(x):
        return np.sin(x * 2 * np.pi + np.pi / 4) + 0.2

    def class_prob(x):
        return std_norm_cdf(true_function(x))

    def obs_function(x):
        fx = class_prob(x)
        return [np.random.uniform() < f for f in fx]

    x_train = np.random.random(20)
    y_train = obs_function(x_train)
    x_test = np.arange(0, 1, 0.01, float)
    gp = GPc.ClassificationGaussianProcess(x_train.reshape(-1, 1), np.asarray(y_train, dtype=int), GPc.ClassifierLikelihood())
    # sweep the length scale, keep the one with the best Laplace evidence
    best = None
    for log_l in np.linspace(-3.0, 0.0, 13):
        lh = np.array([log_l, 0.5])
        nlml = gp.calc_nlml(lh)
        if best is None or nlml < best[0]:
            best = (nlml, lh)
    f, lml = gp.calc_laplace(best[1])
    mean, var, prob = gp.predict(x_test.reshape(-1, 1))
    if verbose:
        print("best log hyp %s  lml %.6f  Newton iterations %d" % (best[1], lml, gp.last_iterations))
        print("mean |P(y=1|x) - true class probability| over the grid: %.3f" % np.mean(np.abs(prob - class_prob(x_test))))
    return dict(lml=lml, log_hyp=best[1], f=f, mean=mean, var=var, prob=prob, truth=class_prob(x_test),
                x_train=x_train, y_train=np.asarray(y_train))


if __name__ == "__main__":
    run()
