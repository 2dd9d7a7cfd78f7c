"""Binary GP classification (Laplace approximation) on the MI355X core -- what the reference's ``GPc.py`` and
``GP_classification_demo.py`` were heading to.

The reference's ``GPc.py`` is an unfinished sketch that does not parse; this module keeps the names it introduces --
``NormCDF`` (GPc.py:4-10), ``logistic_function`` (GPc.py:13-14), ``ClassifierLikelihood(inverse_link_function)`` with
``_preprocess_values`` and ``loglikelihood`` (GPc.py:16-55) -- and adds the model class the demo needs, shaped like the
reference's other Laplace model (``GPpref.PreferenceGaussianProcess``): ``ClassificationGaussianProcess(x_train,
y_train, likelihood, delta_f)`` with ``calc_laplace(loghyp, f=None) -> (f (n,1), lml)``, ``calc_nlml(loghyp)`` and
``predict(x_test) -> (mean, var, prob)``.

What runs where: K (``gpk_fit`` flags = 3: built and kept in HBM, never factored), every Newton iteration of Rasmussen &
Williams Alg. 3.1 (one MFMA Cholesky of B = I + W^1/2 K W^1/2 each) and the prediction of Alg. 3.2: libgpk.so.  The
likelihood classes keep NumPy methods for API use; ``calc_laplace`` does not call them.  PARITY UNPINNED: there is no
runnable reference for this path; tests compare with the dense restatement ``oracle/gpc_oracle.py``.
"""
import ctypes
import warnings

import numpy as np

from . import _lib
from .GPr import fit_handle

__all__ = ["NormCDF", "Logistic", "logistic_function", "ClassifierLikelihood", "ClassificationGaussianProcess"]


def std_norm_cdf(x):
    from scipy.special import ndtr

    return ndtr(x)


def logistic_function(z):
    """GPc.py:13-14."""
    return 1.0 / (1 + np.exp(-z))


class NormCDF(object):
    """Probit likelihood p(y|f) = Phi(y f), y in {-1,+1} (GPc.py:4-10; the reference leaves dlogpyf_df unfinished)."""

    link = 0

    def logpyf(self, y, f):
        return np.log(std_norm_cdf(y * f))

    def dlogpyf_df(self, y, f):
        z = y * f
        return y * np.exp(-0.5 * z * z) / np.sqrt(2 * np.pi) / std_norm_cdf(z)

    def d2logpyf_df2(self, y, f):
        z = y * f
        r = np.exp(-0.5 * z * z) / np.sqrt(2 * np.pi) / std_norm_cdf(z)
        return -(r * r + z * r)


class Logistic(object):
    """Logit likelihood p(y|f) = 1/(1 + exp(-y f)) (the 'Logit' branch of GPc.py:18-19)."""

    link = 1

    def logpyf(self, y, f):
        return -np.log1p(np.exp(-y * f))

    def dlogpyf_df(self, y, f):
        return 0.5 * (y + 1.0) - logistic_function(f)

    def d2logpyf_df2(self, y, f):
        p = logistic_function(f)
        return -p * (1.0 - p)


class ClassifierLikelihood(object):
    """GPc.py:16-55: link selection by name ('Logit' -> logistic, anything else -> probit) and the label check."""

    def __init__(self, inverse_link_function=None):
        if inverse_link_function == "Logit":
            self.inverse_link_function = logistic_function
            self.terms = Logistic()
        else:
            self.inverse_link_function = std_norm_cdf
            self.terms = NormCDF()
        self.Y = None

    def _preprocess_values(self, Y=None):
        """GPc.py:24-38: observations must be in {0,1} or {-1,1}; 0 is mapped to -1."""
        if Y is not None:
            self.Y = np.asarray(Y)
        Y_prep = self.Y.copy().astype(float)
        Y1 = self.Y[self.Y.flatten() == 1].size
        Y2 = self.Y[self.Y.flatten() == 0].size
        Y3 = self.Y[self.Y.flatten() == -1].size
        assert ((Y1 + Y2 == self.Y.size) or (Y1 + Y3 == self.Y.size)), 'Inputs should be in {0,1} or {-1,1}.'
        Y_prep[self.Y.flatten() == 0] = -1
        self.Y = Y_prep
        return Y_prep

    def loglikelihood(self, y, f):
        """log p(y | f), the first term of Psi(f) in Rasmussen eq. 3.12 (GPc.py:40-46)."""
        return self.terms.logpyf(np.asarray(y, dtype=float), np.asarray(f, dtype=float))


class ClassificationGaussianProcess(object):
    """Laplace-approximated binary GP classifier; API shaped like GPpref.PreferenceGaussianProcess (GPpref.py:96-161)."""

    def __init__(self, x_train, y_train, likelihood=None, delta_f=1e-6):
        self.x_train = np.asarray(x_train, dtype=np.float64).reshape(len(x_train), -1)
        self._nx, self._xdim = self.x_train.shape
        self.likelihood = likelihood if likelihood is not None else ClassifierLikelihood()
        self.y_train = self.likelihood._preprocess_values(np.asarray(y_train).reshape(-1))
        if len(self.y_train) != self._nx:
            raise ValueError("y_train has %d entries, x_train %d" % (len(self.y_train), self._nx))
        self.delta_f = delta_f
        self.max_iter = 100
        self.last_iterations = 0
        self.last_trace = None
        self.last_converged = None
        self._mode = None

    def calc_laplace(self, loghyp, f=None):
        """R&W Alg. 3.1 -> (f (n,1), lml).  loghyp = log([l_1..l_d, sigma_f])."""
        loghyp = np.asarray(loghyp, dtype=np.float64).reshape(-1)
        if len(loghyp) != self._xdim + 1:
            raise ValueError("loghyp must have d+1 = %d entries" % (self._xdim + 1))
        lh = np.concatenate([loghyp, [-np.inf]])          # no noise term: log sigma_n = -inf
        handle, _ = fit_handle(self.x_train, None, lh, jitter=0.0, keep_k=True, k_only=True)
        fv = np.zeros(self._nx) if f is None else np.ascontiguousarray(np.asarray(f, dtype=np.float64).reshape(-1)).copy()
        if len(fv) != self._nx:
            raise ValueError("f must have %d entries" % self._nx)
        y = np.ascontiguousarray(self.y_train, dtype=np.float64)
        lml = ctypes.c_double()
        iters = ctypes.c_int()
        trace = np.zeros(self.max_iter)
        rc = _lib.lib().gpk_class_laplace(handle.ptr, y.ctypes.data_as(_lib.c_double_p), int(self.likelihood.terms.link),
                                          fv.ctypes.data_as(_lib.c_double_p), float(self.delta_f), int(self.max_iter),
                                          ctypes.byref(lml), ctypes.byref(iters), trace.ctypes.data_as(_lib.c_double_p))
        _lib.check(rc, "gpk_class_laplace")
        self.last_iterations = iters.value
        self.last_trace = trace[:iters.value].copy()
        self.last_converged = bool(iters.value > 0 and not (self.last_trace[-1] > self.delta_f))
        if not self.last_converged:
            warnings.warn("calc_laplace: Newton loop stopped after max_iter=%d iterations with f_error=%g > delta_f=%g"
                          % (self.max_iter, self.last_trace[-1] if iters.value else float("nan"), self.delta_f),
                          RuntimeWarning)
        self._mode = (handle, y, fv.copy())
        return fv.reshape(self._nx, 1), lml.value

    def calc_nlml(self, loghyp):
        f, lml = self.calc_laplace(loghyp)
        return -lml

    def predict(self, x_test):
        """R&W Alg. 3.2 at the mode of the last calc_laplace -> (latent mean (M,), latent var (M,), P(y=+1) (M,))."""
        if self._mode is None:
            raise RuntimeError("predict needs a calc_laplace first")
        handle, y, fv = self._mode
        Z = np.ascontiguousarray(np.asarray(x_test, dtype=np.float64).reshape(-1, self._xdim))
        mean, mp = _lib.out_array(Z.shape[0])
        var, vp = _lib.out_array(Z.shape[0])
        prob, pp = _lib.out_array(Z.shape[0])
        rc = _lib.lib().gpk_class_predict(handle.ptr, y.ctypes.data_as(_lib.c_double_p), int(self.likelihood.terms.link),
                                          fv.ctypes.data_as(_lib.c_double_p), Z.ctypes.data_as(_lib.c_double_p),
                                          Z.shape[0], mp, vp, pp)
        _lib.check(rc, "gpk_class_predict")
        return mean, var, prob
