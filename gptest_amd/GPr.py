"""Drop-in for the reference's ``GPr.py`` on the MI355X-native core.

Same classes, constructor signatures, attribute names and return shapes as
osurdml/GPtest ``GPr.py`` (file:line cited per member) so that
``GP_regression_demo.py`` runs unchanged with this module on ``sys.path`` as
``GPr`` (see ``dropin/GPr.py`` and INTEGRATION.md).  Every matrix operation is
executed by libgpk.so on the GPU; this module holds no arithmetic beyond
argument marshalling.  Extension over the reference: (N,d) inputs with d > 1
(the reference's squared_distance is 1-D only, GPr.py:5-6).
"""
import ctypes
import hashlib

import numpy as np

from . import _lib

__all__ = ["squared_distance", "GaussianProcess", "MeanFunction", "LikelihoodFunction",
           "CovarianceFunction", "SquaredExponential"]


def _as_points(x):
    """(N,) or (N,1) or (N,d) -> C-contiguous (N,d) float64."""
    a = np.asarray(x, dtype=np.float64)
    if a.ndim == 0:
        a = a.reshape(1, 1)
    elif a.ndim == 1:
        a = a.reshape(len(a), 1)
    elif a.ndim != 2:
        raise ValueError("inputs must be (N,), (N,1) or (N,d)")
    return np.ascontiguousarray(a)


def squared_distance(A, B):
    """GPr.py:4-13 -- pairwise squared distance (N,M); sum over columns for (N,d) inputs."""
    _lib.init()
    A2, B2 = _as_points(A), _as_points(B)
    if A2.shape[1] != B2.shape[1]:
        raise ValueError("dimension mismatch")
    out, outp = _lib.out_array((A2.shape[0], B2.shape[0]))
    rc = _lib.lib().gpk_sqdist(A2.ctypes.data_as(_lib.c_double_p), A2.shape[0],
                               B2.ctypes.data_as(_lib.c_double_p), B2.shape[0], A2.shape[1], outp)
    _lib.check(rc, "gpk_sqdist")
    return out


class MeanFunction(object):
    """GPr.py:72-75."""

    def __init__(self, x):
        self.x = x
        self.y = np.zeros_like(x)


class LikelihoodFunction(object):
    """GPr.py:78-81."""

    def __init__(self, x):
        self.x = x
        self.y = np.exp(2 * x)


class CovarianceFunction(object):
    """GPr.py:84-87."""

    def __init__(self, logHyp, x):
        self.logHyp = logHyp
        self.x = x


class SquaredExponential(CovarianceFunction):
    """GPr.py:90-110; the two matrix builders run on the GPU."""

    def __init__(self, logHyp, x):
        CovarianceFunction.__init__(self, logHyp, x)
        self.hyp = np.exp(self.logHyp)
        n = len(self.hyp)
        self.M = self.hyp[:n - 2]
        self.sf2 = self.hyp[n - 2] ** 2
        self.sn2 = self.hyp[n - 1] ** 2

    def _args(self):
        X = _as_points(self.x)
        lh = np.ascontiguousarray(self.logHyp, dtype=np.float64)
        if len(lh) != X.shape[1] + 2:
            raise ValueError("log hyper-parameters must have d+2 = %d entries, got %d" % (X.shape[1] + 2, len(lh)))
        return X, lh

    def compute_Kxx_matrix(self):
        """GPr.py:99-103."""
        _lib.init()
        X, lh = self._args()
        K, Kp = _lib.out_array((X.shape[0], X.shape[0]))
        rc = _lib.lib().gpk_se_kxx(X.ctypes.data_as(_lib.c_double_p), X.shape[0], X.shape[1],
                                   lh.ctypes.data_as(_lib.c_double_p), 0.0, Kp)
        _lib.check(rc, "gpk_se_kxx")
        return K

    def compute_Kxz_matrix(self, z):
        """GPr.py:105-110."""
        _lib.init()
        X, lh = self._args()
        Z = _as_points(z)
        if Z.shape[1] != X.shape[1]:
            raise ValueError("test inputs have dimension %d, training inputs %d" % (Z.shape[1], X.shape[1]))
        K, Kp = _lib.out_array((X.shape[0], Z.shape[0]))
        rc = _lib.lib().gpk_se_kxz(X.ctypes.data_as(_lib.c_double_p), X.shape[0], Z.ctypes.data_as(_lib.c_double_p),
                                   Z.shape[0], X.shape[1], lh.ctypes.data_as(_lib.c_double_p), Kp)
        _lib.check(rc, "gpk_se_kxz")
        return K


class _Handle(object):
    """Owns one gpk_model; freed with the Python object."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().gpk_free(self.ptr)
        except Exception:
            pass
        self.ptr = None


def fit_handle(X, y, log_hyp, jitter=0.0, keep_k=False, k_only=False):
    """gpk_fit -> (_Handle, nlml).  Raises numpy.linalg.LinAlgError if K is not positive definite.
    ``k_only`` (with ``keep_k`` and ``y=None``): K is built and kept in HBM, nothing is factored."""
    _lib.init()
    X = _as_points(X)
    lh = np.ascontiguousarray(log_hyp, dtype=np.float64)
    if len(lh) != X.shape[1] + 2:
        raise ValueError("log hyper-parameters must have d+2 = %d entries, got %d" % (X.shape[1] + 2, len(lh)))
    yp = None
    if y is not None:
        yv = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        if len(yv) != X.shape[0]:
            raise ValueError("targets have %d entries, inputs %d" % (len(yv), X.shape[0]))
        yp = yv.ctypes.data_as(_lib.c_double_p)
    h = ctypes.c_void_p()
    nlml = ctypes.c_double()
    rc = _lib.lib().gpk_fit(X.ctypes.data_as(_lib.c_double_p), yp, X.shape[0], X.shape[1],
                            lh.ctypes.data_as(_lib.c_double_p), float(jitter), (1 if keep_k else 0) | (2 if k_only else 0),
                            ctypes.byref(h), ctypes.byref(nlml))
    _lib.check(rc, "gpk_fit")
    return _Handle(h), nlml.value


class GaussianProcess(object):
    """GPr.py:16-69."""

    def __init__(self, log_hyp, mean_hyp, like_hyp, covFunName, meanFunName, likeFunName,
                 trainInput, trainTarget):
        self.log_hyp = log_hyp
        self.mean_hyp = mean_hyp
        self.like_hyp = like_hyp
        self.covFunName = covFunName
        self.meanFunName = meanFunName
        self.likeFunName = likeFunName
        self.trainInput = trainInput
        self.trainTarget = trainTarget
        # GPr.py:29-42: selection by string name; unknown names yield []
        self.covFun = SquaredExponential(log_hyp, trainInput) if covFunName == "SE" else []
        self.meanFun = MeanFunction(mean_hyp) if meanFunName == "zero" else []
        self.likeFun = LikelihoodFunction(like_hyp) if likeFunName == "zero" else []
        self._cache_key = None
        self._cache = None

    def _key(self, lh):
        """Cache key of a factorisation: the hyper-parameters AND the training data as they are now.  The reference
        re-reads self.trainInput / self.trainTarget on every call (GPr.py:47,59-61), so reassigning or mutating them
        in place must invalidate the cached factor: a digest of the bytes (1 MiB at N = 16384, d = 8: < 1 ms)."""
        h = hashlib.blake2b(digest_size=16)
        h.update(np.ascontiguousarray(lh, dtype=np.float64).tobytes())
        for a in (self.trainInput, self.trainTarget):
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
            h.update(repr(a.shape).encode())
            h.update(a.tobytes())
        return h.digest()

    def _fitted(self, hyp):
        """One K build + one factorisation per distinct (hyper-parameters, data) (the reference
        rebuilds and refactors on every call, GPr.py:47-48,59-63; results are identical)."""
        lh = np.ascontiguousarray(hyp, dtype=np.float64)
        key = self._key(lh)
        if self._cache_key != key:
            self._cache = fit_handle(self.trainInput, self.trainTarget, lh)
            self._cache_key = key
        return self._cache

    def compute_prediction(self, testInput):
        """GPr.py:45-54 -> (fz (M,), cov_fz (M,)); latent variance, no +sn2 (GPr.py:51)."""
        Z = _as_points(testInput)
        X = _as_points(self.trainInput)
        if Z.shape[1] != X.shape[1]:
            raise ValueError("test inputs have dimension %d, training inputs %d" % (Z.shape[1], X.shape[1]))
        fz, fzp = _lib.out_array(Z.shape[0])
        cov, covp = _lib.out_array(Z.shape[0])
        lh = np.ascontiguousarray(self.covFun.logHyp, dtype=np.float64)
        key = self._key(lh)
        if self._cache_key != key:
            # nothing factored yet for these hyper-parameters: one fused call (K build, Cholesky and the
            # prediction solve pipelined behind it), and the handle is kept for later calls
            _lib.init()
            if len(lh) != X.shape[1] + 2:
                raise ValueError("log hyper-parameters must have d+2 = %d entries, got %d" % (X.shape[1] + 2, len(lh)))
            yv = np.ascontiguousarray(np.asarray(self.trainTarget, dtype=np.float64).reshape(-1))
            if len(yv) != X.shape[0]:
                raise ValueError("targets have %d entries, inputs %d" % (len(yv), X.shape[0]))
            h = ctypes.c_void_p()
            nlml = ctypes.c_double()
            rc = _lib.lib().gpk_fit_predict(X.ctypes.data_as(_lib.c_double_p), yv.ctypes.data_as(_lib.c_double_p),
                                            X.shape[0], X.shape[1], lh.ctypes.data_as(_lib.c_double_p), 0.0,
                                            Z.ctypes.data_as(_lib.c_double_p), Z.shape[0], ctypes.byref(h),
                                            ctypes.byref(nlml), fzp, covp)
            _lib.check(rc, "gpk_fit_predict")
            self._cache = (_Handle(h), nlml.value)
            self._cache_key = key
            return fz, cov
        handle, _ = self._fitted(self.covFun.logHyp)
        rc = _lib.lib().gpk_predict(handle.ptr, Z.ctypes.data_as(_lib.c_double_p), Z.shape[0], fzp, covp)
        _lib.check(rc, "gpk_predict")
        return fz, cov

    def compute_likelihood(self, hyp):
        """GPr.py:57-69 -> (1,1) ndarray; uses ``hyp``, not self.log_hyp (GPr.py:59)."""
        if self.meanFun == [] or self.covFunName != "SE":
            raise AttributeError("compute_likelihood needs covFunName='SE' and meanFunName='zero'")
        _, nlml = self._fitted(hyp)
        return np.array([[nlml]])

    def compute_likelihood_gradient(self, hyp):
        """Extension (not in the reference, whose demos hand-set ``hyp``): ``(nlml (1,1), grad (d+2,))`` with
        ``grad = d nlml / d hyp`` in the log parametrisation of GPr.py:93-97, from the factor
        ``compute_likelihood(hyp)`` already holds in HBM.  Suits ``scipy.optimize.minimize(..., jac=True)``."""
        if self.meanFun == [] or self.covFunName != "SE":
            raise AttributeError("compute_likelihood_gradient needs covFunName='SE' and meanFunName='zero'")
        handle, nlml = self._fitted(hyp)
        grad, gp = _lib.out_array(len(np.asarray(hyp).reshape(-1)))
        _lib.check(_lib.lib().gpk_nlml_grad(handle.ptr, gp), "gpk_nlml_grad")
        return np.array([[nlml]]), grad
