"""Drop-in for the reference's ``GPpref.py`` (preference GP, Laplace approximation) on the MI355X core.

Same public surface as osurdml/GPtest ``GPpref.py``: ``PrefProbit`` (GPpref.py:46-94),
``PreferenceGaussianProcess(x_train, uvi_train, y_train, likelihood=PrefProbit, delta_f=1e-6)`` with
``calc_laplace(loghyp, f=None) -> (f (nx,1), lml)`` and ``calc_nlml(loghyp)`` (GPpref.py:96-161),
``squared_distance`` (GPpref.py:13-22), ``std_norm_pdf`` (GPpref.py:8-10).  Runs under Python 3 and does
not need GPy: the kernel matrix ``GPy.kern.RBF(ARD=True).K`` (call sites GPpref.py:109,122) is the same
``variance * exp(-0.5 * sum_k ((x_ik - x_jk)/l_k)^2)`` that ``gpk_fit`` builds on the GPU.

What runs where
* K + eps*I, its Cholesky, logdetK and every Newton iteration (GPpref.py:122-155): libgpk.so
  (``gpk_fit`` with the keep-K flag, then ``gpk_pref_laplace``).  The reference forms the explicit inverse
  ``iK`` and ``inv(iK + W)`` (2 nx^3 per iteration); the GPU path never inverts K (see k_pref.hip).
* ``PrefProbit`` keeps the reference's methods and return values (``derivatives`` still returns the dense W), with the
  per-pair terms computed by the same device kernel as the Laplace loop (``gpk_pref_pair_stats``); ``calc_laplace`` itself
  never forms W.

Reference quirks kept for parity: probit sigma stays at the constructor default because
``self.likelihood.set_sigma = ...`` (GPpref.py:115) assigns over the method; ``logdetK`` is
``sum(log(diag L))`` (GPpref.py:131) and ``log_marginal`` halves it again (GPpref.py:93); the Cholesky is
retried with 10x jitter on ``LinAlgError`` (GPpref.py:126-135).  Parity for this module is UNPINNED (the
reference cannot be imported: Python-2 + GPy); the tests compare against a NumPy restatement of GPpref.py.
"""
import ctypes
import warnings

import numpy as np

from . import _lib
from .GPr import fit_handle, squared_distance  # noqa: F401  (squared_distance: GPpref.py:13-22)

__all__ = ["std_norm_pdf", "squared_distance", "PrefProbit", "PreferenceGaussianProcess"]

_sqrt_2pi = np.sqrt(2 * np.pi)


def std_norm_cdf(x):
    from scipy.special import ndtr

    return ndtr(x)


def std_norm_pdf(x):
    """GPpref.py:8-10."""
    x = np.clip(x, -1e150, 1e150)
    return np.exp(-(x ** 2) / 2) / _sqrt_2pi


class PrefProbit(object):
    """The probit pair likelihood, GPpref.py:46-94: same methods, same return values (``derivatives`` still hands back
    the dense ``W`` and the column ``dpy_df`` the reference's callers expect), but the per-pair terms -- ``Phi``, ``N``,
    their ratio, the gradient and curvature weights, ``log Phi`` -- come from the device kernel the Laplace loop itself
    runs (``pref_pair_stats_kernel`` through ``gpk_pref_pair_stats``); only the scatter into the caller's dense arrays is
    NumPy.  The reference's scatter rules are kept: ``dpy_df[idx] += v`` drops repeated indices (GPpref.py:77-78) while
    the ``W`` loop accumulates them (GPpref.py:82-87)."""

    def __init__(self, sigma=1.0):
        self.set_sigma(sigma)
        self.log2pi = np.log(2.0 * np.pi)

    def set_sigma(self, sigma):
        self.sigma = sigma
        self._isqrt2sig = 1.0 / (self.sigma * np.sqrt(2.0))
        self._i2var = self._isqrt2sig ** 2

    def _pair_terms(self, uvi, y, f):
        """(z, q, c, log Phi) per pair from the device."""
        uv = np.ascontiguousarray(uvi, dtype=np.int64)
        npairs = uv.shape[0]
        fv, fp = _lib.as_c(np.reshape(f, -1))
        yv, yp = _lib.as_c(np.broadcast_to(np.reshape(y, -1), (npairs,)))
        out, op = _lib.out_array((4, npairs))
        ptr = lambda k: ctypes.cast(ctypes.addressof(op.contents) + 8 * k * npairs, _lib.c_double_p)
        _lib.init()
        _lib.check(_lib.lib().gpk_pref_pair_stats(uv.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), yp, npairs,
                                                  float(self.sigma), fp, len(fv), ptr(0), ptr(1), ptr(2), ptr(3)),
                   "gpk_pref_pair_stats")
        return out[0], out[1], out[2], out[3]

    def z_k(self, uvi, f, y):
        zc = self._isqrt2sig * (f[uvi[:, 1]] - f[uvi[:, 0]])
        return y * zc

    def I_k(self, x, uv):
        return -1 if x == uv[0] else (1 if x == uv[1] else 0)

    def derivatives(self, uvi, y, f):
        """(W (nx, nx), dpy_df (nx, 1)) as GPpref.py:68-88 returns them."""
        nx = len(f)
        _, q, c, _ = self._pair_terms(uvi, y, f)
        u, v = np.asarray(uvi)[:, 0], np.asarray(uvi)[:, 1]
        dpy_df = np.zeros((nx, 1), dtype="float")
        dpy_df[u] += -q.reshape(-1, 1)
        dpy_df[v] += q.reshape(-1, 1)
        W = np.zeros((nx, nx), dtype="float")
        np.add.at(W, (u, u), c)
        np.add.at(W, (v, v), c)
        np.subtract.at(W, (u, v), c)
        np.subtract.at(W, (v, u), c)
        return W, dpy_df

    def log_marginal(self, uvi, y, f, iK, logdetK):
        _, _, _, logphi = self._pair_terms(uvi, y, f)
        psi = (np.sum(logphi) - 0.5 * np.matmul(np.matmul(f.T, iK), f)
               - 0.5 * logdetK - iK.shape[0] / 2.0 * self.log2pi)
        return psi.flat[0]


class _RBFParams(object):
    """Stand-in for the two GPy kernel attributes the reference sets (GPpref.py:113-114)."""

    def __init__(self, dim):
        self.lengthscale = np.ones(dim)
        self.variance = 1.0


class PreferenceGaussianProcess(object):
    """GPpref.py:96-161."""

    def __init__(self, x_train, uvi_train, y_train, likelihood=PrefProbit, delta_f=1e-6):
        self._xdim = x_train.shape[1]
        self._nx = x_train.shape[0]
        self.x_train = x_train
        self.y_train = y_train
        self.uvi_train = uvi_train
        self.delta_f = delta_f
        self.likelihood = likelihood()
        self.kern = _RBFParams(self._xdim)
        self.max_iter = 100      # the reference loops without a bound (GPpref.py:140); see last_converged
        self.last_converged = None
        self.verbose = False     # the reference prints f_error, lml every iteration (GPpref.py:154)
        self.last_iterations = 0
        self.last_eps = None
        self.last_trace = None
        self._mode = None

    def calc_laplace(self, loghyp, f=None):
        """GPpref.py:112-157 -> (f (nx,1), lml)."""
        loghyp = np.asarray(loghyp, dtype=np.float64)
        self.kern.lengthscale = np.exp(loghyp[0:self._xdim])
        self.kern.variance = (np.exp(loghyp[self._xdim])) ** 2
        self.likelihood.set_sigma = np.exp(loghyp[-1])  # GPpref.py:115, verbatim: sigma itself is unchanged
        if f is None:
            f = np.zeros((self._nx, 1))
        # K = variance * exp(-r^2/2), no noise term: log sigma_n = -inf; jitter eps on the diagonal
        lh = np.concatenate([loghyp[0:self._xdim], [loghyp[self._xdim]], [-np.inf]])
        eps = 1e-6
        # GPpref.py:126-135 retries with 10x the jitter for ever.  K + eps I with |K_ij| <= variance is positive
        # definite for certain once eps > nx * variance; a failure beyond that means the inputs are not finite
        # (NaN in x_train makes every retry fail), so the loop is bounded there and raises.
        eps_cap = 10.0 * max(1.0, float(self._nx) * float(self.kern.variance))
        while True:
            try:
                handle, _ = fit_handle(self.x_train, None, lh, jitter=eps, keep_k=True)
                break
            except np.linalg.LinAlgError:
                eps = eps * 10
                if self.verbose:
                    print("Inversion issue, adding noise: {0}".format(eps))
                if not (eps <= eps_cap):
                    raise np.linalg.LinAlgError(
                        "K + eps*I is not positive definite even with eps = %g: non-finite inputs or "
                        "hyper-parameters?" % eps)
        self.last_eps = eps
        uvi = np.ascontiguousarray(self.uvi_train, dtype=np.int64)
        y = np.ascontiguousarray(np.asarray(self.y_train, dtype=np.float64).reshape(-1))
        fv = np.ascontiguousarray(np.asarray(f, dtype=np.float64).reshape(-1)).copy()
        if uvi.shape != (len(y), 2) or len(fv) != self._nx:
            raise ValueError("uvi_train must be (npairs,2), y_train (npairs,1), f (nx,1)")
        lml = ctypes.c_double()
        iters = ctypes.c_int()
        trace = np.zeros(self.max_iter)
        rc = _lib.lib().gpk_pref_laplace(handle.ptr, uvi.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
                                         y.ctypes.data_as(_lib.c_double_p), len(y), float(self.likelihood.sigma),
                                         fv.ctypes.data_as(_lib.c_double_p), float(self.delta_f), int(self.max_iter),
                                         ctypes.byref(lml), ctypes.byref(iters), trace.ctypes.data_as(_lib.c_double_p))
        _lib.check(rc, "gpk_pref_laplace")
        self.last_iterations = iters.value
        self.last_trace = trace[:iters.value].copy()
        # the reference iterates until f_error <= delta_f whatever it takes (GPpref.py:140); this loop stops at
        # max_iter and says so instead of returning a non-converged mode silently
        self.last_converged = bool(iters.value > 0 and not (self.last_trace[-1] > self.delta_f))
        if not self.last_converged:
            warnings.warn("calc_laplace: Newton loop stopped after max_iter=%d iterations with f_error=%g > "
                          "delta_f=%g" % (self.max_iter, self.last_trace[-1] if iters.value else float("nan"),
                                          self.delta_f), RuntimeWarning)
        self._mode = (handle, uvi, y, fv.copy())   # kept in HBM / on the host for predict_latent
        if self.verbose:
            for e in self.last_trace:
                print(e)
        return fv.reshape(self._nx, 1), lml.value

    def predict_latent(self, x_test):
        """Extension (the reference stops at the mode, GPpref.py:157): latent posterior at ``x_test`` (M, d)
        under the Laplace approximation of the last ``calc_laplace`` -> (mean (M,), var (M,)).
        ``mean = k*^T (K+eps I)^-1 f``, ``var = k** - k*^T (K + eps I + W^-1)^-1 k*`` (R&W eq. 3.21, 3.24)."""
        if getattr(self, "_mode", None) is None:
            raise RuntimeError("predict_latent needs a calc_laplace first")
        handle, uvi, y, fv = self._mode
        Z = np.ascontiguousarray(np.asarray(x_test, dtype=np.float64).reshape(-1, self._xdim))
        mean, mp = _lib.out_array(Z.shape[0])
        var, vp = _lib.out_array(Z.shape[0])
        rc = _lib.lib().gpk_pref_predict(handle.ptr, uvi.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
                                         y.ctypes.data_as(_lib.c_double_p), len(y), float(self.likelihood.sigma),
                                         fv.ctypes.data_as(_lib.c_double_p), Z.ctypes.data_as(_lib.c_double_p),
                                         Z.shape[0], mp, vp)
        _lib.check(rc, "gpk_pref_predict")
        return mean, var

    def calc_nlml(self, loghyp):
        """GPpref.py:159-161."""
        f, lml = self.calc_laplace(loghyp)
        return -lml
