"""The handful of ``GPy`` calls the reference's scripts make, on the MI355X core.

``GP_parameter_fit.py`` drives a third-party ``GPy`` model: ``GPy.kern.RBF(input_dim=2, variance=10., lengthscale=20.)``
(:30), ``GPy.models.GPRegression(X, Y, kernel)`` (:31), ``gpm.optimize(...)`` / ``optimize_restarts`` (:32-33),
``model.predict(Xfull) -> (mean, var)`` (:52), ``model.X`` (:56) and, once per video frame, ``gpm.set_XY(X[0:ii+1],
Y[0:ii+1] - mean_value)`` (:62): a data set that grows by one observation at a time.  GPy refactors all of K on every
``set_XY``; here a ``set_XY`` whose data extend the current ones goes through ``gpk_append`` (the factor in HBM is
extended by the new rows, O(n^2) per point), anything else refits.  ``GPpref.py`` uses ``GPy.kern.RBF(d, ARD=True)`` with
``.lengthscale``, ``.variance`` and ``.K(x)`` (GPpref.py:109,113-114,122), also provided.

GPy itself is absent from this image and un-pinned by the reference: PARITY UNPINNED; semantics follow GPy's documented
behaviour (RBF: variance * exp(-r^2/2) with r the lengthscale-scaled distance; Gaussian likelihood variance 1.0 by default;
``predict`` returns the variance of y, i.e. latent variance + noise variance; outputs shaped (M, 1)).
``dropin/GPy`` exposes these classes as ``GPy.kern.RBF`` / ``GPy.models.GPRegression``.
"""
import ctypes

import numpy as np

from . import _lib
from .GPr import _Handle, _as_points

__all__ = ["RBF", "GPRegression"]


class RBF(object):
    """GPy.kern.RBF(input_dim, variance=1., lengthscale=None, ARD=False)."""

    def __init__(self, input_dim, variance=1.0, lengthscale=None, ARD=False):
        self.input_dim = int(input_dim)
        self.ARD = bool(ARD)
        self.variance = float(variance)
        n = self.input_dim if self.ARD else 1
        if lengthscale is None:
            lengthscale = np.ones(n)
        self.lengthscale = np.asarray(lengthscale, dtype=np.float64).reshape(-1) * np.ones(n)

    def _log_hyp(self, log_sn=-np.inf):
        ls = np.asarray(self.lengthscale, dtype=np.float64).reshape(-1)
        ell = ls if len(ls) == self.input_dim else np.full(self.input_dim, ls[0])
        return np.concatenate([np.log(ell), [0.5 * np.log(float(np.asarray(self.variance).reshape(-1)[0]))], [log_sn]])

    def K(self, X, X2=None):
        """variance * exp(-0.5 * sum_k ((x_ik - x_jk) / l_k)^2) -- built on the GPU (gpk_se_kxx / gpk_se_kxz)."""
        _lib.init()
        A = _as_points(X)
        lh = np.ascontiguousarray(self._log_hyp())
        if X2 is None:
            out, op = _lib.out_array((A.shape[0], A.shape[0]))
            rc = _lib.lib().gpk_se_kxx(A.ctypes.data_as(_lib.c_double_p), A.shape[0], A.shape[1],
                                       lh.ctypes.data_as(_lib.c_double_p), 0.0, op)
            _lib.check(rc, "gpk_se_kxx")
            return out
        B = _as_points(X2)
        out, op = _lib.out_array((A.shape[0], B.shape[0]))
        rc = _lib.lib().gpk_se_kxz(A.ctypes.data_as(_lib.c_double_p), A.shape[0], B.ctypes.data_as(_lib.c_double_p),
                                   B.shape[0], A.shape[1], lh.ctypes.data_as(_lib.c_double_p), op)
        _lib.check(rc, "gpk_se_kxz")
        return out


class GPRegression(object):
    """GPy.models.GPRegression(X, Y, kernel=None, noise_var=1.) -- zero mean, Gaussian likelihood."""

    def __init__(self, X, Y, kernel=None, noise_var=1.0):
        self.X = np.array(_as_points(X))
        self.Y = np.asarray(Y, dtype=np.float64).reshape(-1, 1).copy()
        if len(self.Y) != len(self.X):
            raise ValueError("X has %d rows, Y %d" % (len(self.X), len(self.Y)))
        self.kern = kernel if kernel is not None else RBF(self.X.shape[1])
        self.noise_var = float(noise_var)
        self._handle = None
        self._nlml = None
        self._fit_key = None
        self.appended_points = 0   # observations that entered through gpk_append instead of a refactorisation
        self.refits = 0

    # ---- internals
    def _log_hyp(self):
        return np.ascontiguousarray(self.kern._log_hyp(0.5 * np.log(self.noise_var)))

    def _ensure_fit(self):
        key = self._log_hyp().tobytes()
        if self._handle is not None and self._fit_key == key:
            return
        _lib.init()
        lh = self._log_hyp()
        X = np.ascontiguousarray(self.X)
        y = np.ascontiguousarray(self.Y.reshape(-1))
        h = ctypes.c_void_p()
        nlml = ctypes.c_double()
        rc = _lib.lib().gpk_fit(X.ctypes.data_as(_lib.c_double_p), y.ctypes.data_as(_lib.c_double_p), X.shape[0], X.shape[1],
                                lh.ctypes.data_as(_lib.c_double_p), 0.0, 4, ctypes.byref(h), ctypes.byref(nlml))
        _lib.check(rc, "gpk_fit")
        self._handle = _Handle(h)
        self._nlml = nlml.value
        self._fit_key = key
        self.refits += 1

    # ---- GPy surface
    def log_likelihood(self):
        self._ensure_fit()
        return -self._nlml

    def predict(self, Xnew, full_cov=False, include_likelihood=True):
        """-> (mean (M,1), var (M,1)); the variance of y (latent + noise) unless include_likelihood=False."""
        if full_cov:
            raise NotImplementedError("full_cov=True is not provided")
        self._ensure_fit()
        Z = _as_points(Xnew)
        mean, mp = _lib.out_array(Z.shape[0])
        var, vp = _lib.out_array(Z.shape[0])
        rc = _lib.lib().gpk_predict(self._handle.ptr, Z.ctypes.data_as(_lib.c_double_p), Z.shape[0], mp, vp)
        _lib.check(rc, "gpk_predict")
        if include_likelihood:
            var = var + self.noise_var
        return mean.reshape(-1, 1), var.reshape(-1, 1)

    def set_XY(self, X=None, Y=None):
        """GPy's set_XY.  Data that EXTEND the current ones (same leading rows) are appended to the factor held in
        HBM (gpk_append); anything else invalidates it."""
        Xn = self.X if X is None else np.array(_as_points(X))
        Yn = self.Y if Y is None else np.asarray(Y, dtype=np.float64).reshape(-1, 1).copy()
        if len(Xn) != len(Yn):
            raise ValueError("X has %d rows, Y %d" % (len(Xn), len(Yn)))
        n = len(self.X)
        extends = (self._handle is not None and self._fit_key == self._log_hyp().tobytes() and len(Xn) > n
                   and Xn.shape[1] == self.X.shape[1] and np.array_equal(Xn[:n], self.X) and np.array_equal(Yn[:n], self.Y))
        if extends:
            xt = np.ascontiguousarray(Xn[n:])
            yt = np.ascontiguousarray(Yn[n:].reshape(-1))
            hp = ctypes.c_void_p(self._handle.ptr.value)
            nlml = ctypes.c_double()
            rc = _lib.lib().gpk_append(ctypes.byref(hp), xt.ctypes.data_as(_lib.c_double_p), yt.ctypes.data_as(_lib.c_double_p),
                                       len(xt), ctypes.byref(nlml))
            if rc == 0:
                self._handle.ptr = hp          # gpk_append may have replaced (and freed) the old model
                self._nlml = nlml.value
                self.appended_points += len(xt)
            else:
                if hp.value == self._handle.ptr.value:
                    pass                       # extended in place and failed: the handle only remains to be freed
                self._handle = None
                self._fit_key = None
                self.X, self.Y = Xn, Yn
                _lib.check(rc, "gpk_append")
        elif not (len(Xn) == n and np.array_equal(Xn, self.X) and np.array_equal(Yn, self.Y)):
            self._handle = None
            self._fit_key = None
        self.X, self.Y = Xn, Yn

    def _objective(self, theta):
        """nlml and its gradient in theta = (log variance, log lengthscale(s), log noise variance)."""
        nl = 1 if not self.kern.ARD else self.kern.input_dim
        self.kern.variance = float(np.exp(theta[0]))
        self.kern.lengthscale = np.exp(theta[1:1 + nl])
        self.noise_var = float(np.exp(theta[1 + nl]))
        self._ensure_fit()
        d = self.X.shape[1]
        g, gp = _lib.out_array(d + 2)
        _lib.check(_lib.lib().gpk_nlml_grad(self._handle.ptr, gp), "gpk_nlml_grad")
        # chain rule: log_hyp = [log l_k, log sigma_f = theta0 / 2, log sigma_n = theta_noise / 2]
        gl = g[:d] if self.kern.ARD else np.array([g[:d].sum()])
        return self._nlml, np.concatenate([[0.5 * g[d]], gl, [0.5 * g[d + 1]]])

    def _theta(self):
        return np.concatenate([[np.log(self.kern.variance)], np.log(np.asarray(self.kern.lengthscale).reshape(-1)),
                               [np.log(self.noise_var)]])

    def optimize(self, optimizer=None, messages=False, max_iters=1000, start=None):
        """Maximise the log marginal likelihood over (variance, lengthscale(s), noise variance) with L-BFGS-B, using
        the analytic gradient computed on the GPU."""
        from scipy.optimize import minimize

        t0 = self._theta() if start is None else np.asarray(start, dtype=float)
        best = minimize(self._objective, t0, jac=True, method="L-BFGS-B", bounds=[(-12.0, 12.0)] * len(t0),
                        options=dict(maxiter=int(max_iters)))
        self._objective(best.x)
        if messages:
            print("optimize: nlml %.6f after %d evaluations" % (best.fun, best.nfev))
        return best

    def optimize_restarts(self, num_restarts=10, robust=True, verbose=False, **kw):
        """Random restarts (theta ~ N(0,1) around the current optimum, as GPy randomises), the best one is kept."""
        rng = np.random.RandomState(0)
        best_t, best_f = self._theta(), -self.log_likelihood()
        for _ in range(int(num_restarts)):
            try:
                r = self.optimize(start=best_t * 0 + rng.randn(len(best_t)), **kw)
            except (np.linalg.LinAlgError, _lib.GpkError):
                if not robust:
                    raise
                continue
            if r.fun < best_f:
                best_t, best_f = r.x.copy(), r.fun
            if verbose:
                print("restart: nlml %.6f" % r.fun)
        self._objective(best_t)
        return best_f

    def __str__(self):
        return ("Name : GP regression\nObjective : %.6f\nrbf.variance %s\nrbf.lengthscale %s\nGaussian_noise.variance %s"
                % (-self.log_likelihood(), self.kern.variance, self.kern.lengthscale, self.noise_var))

    __repr__ = __str__
