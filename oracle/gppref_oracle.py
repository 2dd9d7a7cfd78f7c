"""NumPy restatement of the reference preference-GP Laplace path (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED.  The reference's ``GPpref.py`` cannot be imported here: it is
Python-2 source (``print`` statements GPpref.py:135,154) and its kernel matrix
comes from the third-party ``GPy`` (``GPy.kern.RBF(input_dim, ARD=True).K``,
call sites GPpref.py:109,113-114,122), which is not installed, not vendored and
not version-pinned by the reference.  No reference test or fixture pins this
path.  This file restates the text of GPpref.py:46-94 (likelihood) and
GPpref.py:112-157 (Laplace loop) with ``kern.K`` restated from GPy's documented
RBF-ARD definition ``variance * exp(-0.5 * sum_k ((x_ik - x_jk)/l_k)^2)``.
Fixtures produced from it (``tests/golden/gppref_*.npz``) are labelled
"restatement-derived, not reference-derived".

Reference quirks reproduced on purpose:
* probit sigma stays 1.0: ``self.likelihood.set_sigma = ...`` (GPpref.py:115)
  assigns over the method instead of calling it;
* ``logdetK`` is sum(log(diag L)) = 0.5 log|K| (GPpref.py:131) and
  ``log_marginal`` halves it again (GPpref.py:93);
* the gradient scatter ``dpy_df[idx] += v`` (GPpref.py:77-78) drops duplicate
  indices while the W loop (GPpref.py:82-87) accumulates them.
"""
import numpy as np
from scipy.special import ndtr as std_norm_cdf

_sqrt_2pi = np.sqrt(2 * np.pi)


def std_norm_pdf(x):
    """GPpref.py:8-10."""
    x = np.clip(x, -1e150, 1e150)
    return np.exp(-(x ** 2) / 2) / _sqrt_2pi


def rbf_ard_K(x, lengthscale, variance):
    """GPy.kern.RBF(ARD=True).K(X) restated (call site GPpref.py:122); no noise term."""
    xs = np.asarray(x, dtype=float) / np.asarray(lengthscale, dtype=float)
    d2 = np.zeros((xs.shape[0], xs.shape[0]))
    for k in range(xs.shape[1]):
        diff = xs[:, k : k + 1] - xs[:, k : k + 1].T
        d2 += diff * diff
    return variance * np.exp(-0.5 * d2)


class PrefProbit(object):
    """GPpref.py:46-94."""

    def __init__(self, sigma=1.0):
        self.set_sigma(sigma)
        self.log2pi = np.log(2.0 * np.pi)

    def set_sigma(self, sigma):
        self.sigma = sigma
        self._isqrt2sig = 1.0 / (self.sigma * np.sqrt(2.0))
        self._i2var = self._isqrt2sig ** 2

    def z_k(self, uvi, f, y):
        zc = self._isqrt2sig * (f[uvi[:, 1]] - f[uvi[:, 0]])
        return y * zc

    def derivatives(self, uvi, y, f):
        nx = len(f)
        z = self.z_k(uvi, f, y)
        phi_z = std_norm_cdf(z)
        N_z = std_norm_pdf(z)
        dpy_df = np.zeros((nx, 1), dtype="float")
        dpyuv_df = y * self._isqrt2sig * N_z / phi_z
        dpy_df[uvi[:, 0]] += -dpyuv_df  # duplicates NOT accumulated (GPpref.py:77)
        dpy_df[uvi[:, 1]] += dpyuv_df
        inner = -self._i2var * (z * N_z / phi_z + (N_z / phi_z) ** 2)
        W = np.zeros((nx, nx), dtype="float")
        for uvik, ddpy_df in zip(uvi, inner):
            xi, yi = uvik
            ddpy_df = ddpy_df.item() if hasattr(ddpy_df, 'item') else ddpy_df
            W[xi, xi] -= ddpy_df
            W[yi, yi] -= ddpy_df
            W[xi, yi] -= -ddpy_df
            W[yi, xi] -= -ddpy_df
        return W, dpy_df

    def log_marginal(self, uvi, y, f, iK, logdetK):
        z = self.z_k(uvi, f, y)
        phi_z = std_norm_cdf(z)
        psi = (np.sum(np.log(phi_z)) - 0.5 * np.matmul(np.matmul(f.T, iK), f)
               - 0.5 * logdetK - iK.shape[0] / 2.0 * self.log2pi)
        return psi.flat[0]


def calc_laplace(x_train, uvi_train, y_train, loghyp, f=None, delta_f=1e-6, max_iter=100, trace=None):
    """GPpref.py:112-157.  Returns (f (nx,1), lml, n_iter, eps)."""
    xdim = x_train.shape[1]
    nx = x_train.shape[0]
    lik = PrefProbit()  # sigma stays 1.0 (GPpref.py:115)
    lengthscale = np.exp(loghyp[0:xdim])
    variance = (np.exp(loghyp[xdim])) ** 2
    if f is None:
        f = np.zeros((nx, 1))
    Ix = np.eye(nx)
    K = rbf_ard_K(x_train, lengthscale, variance)
    eps = 1e-6
    while True:
        try:
            L = np.linalg.cholesky(K + eps * Ix)
            iK = np.linalg.solve(L.T, np.linalg.solve(L, Ix))
            logdetK = np.sum(np.log(L.diagonal()))
            break
        except np.linalg.LinAlgError:
            eps = eps * 10
    f_error = delta_f + 1
    it = 0
    lml = None
    while f_error > delta_f and it < max_iter:
        W, dpy_df = lik.derivatives(uvi_train, y_train, f)
        g = iK + W
        f_new = np.matmul(np.linalg.inv(g), np.matmul(W, f) + dpy_df)
        lml = lik.log_marginal(uvi_train, y_train, f_new, iK, logdetK)
        f_error = np.max(np.abs(f_new - f))
        if trace is not None:
            trace.append((f_error, lml))
        f = f_new
        it += 1
    return f, lml, it, eps


def predict_latent(x_train, uvi_train, y_train, loghyp, f_hat, eps, x_test):
    """Latent Laplace posterior at x_test (Rasmussen & Williams eq. 3.21 / 3.24); the reference stops at the mode
    (GPpref.py:157), so this row has no reference output at all: PARITY UNPINNED, dense textbook algebra.

    mean = k*^T Ke^-1 f,  var = k** - k*^T (I + W Ke)^-1 W k*   (= k** - k*^T (Ke + W^-1)^-1 k*, W singular).
    """
    xdim = x_train.shape[1]
    nx = x_train.shape[0]
    lik = PrefProbit()
    lengthscale = np.exp(loghyp[0:xdim])
    variance = (np.exp(loghyp[xdim])) ** 2
    Ke = rbf_ard_K(x_train, lengthscale, variance) + eps * np.eye(nx)
    xs = np.asarray(x_train, dtype=float) / lengthscale
    zs = np.asarray(x_test, dtype=float).reshape(-1, xdim) / lengthscale
    d2 = ((xs[:, None, :] - zs[None, :, :]) ** 2).sum(axis=2)
    Ks = variance * np.exp(-0.5 * d2)                      # (nx, M)
    W, _ = lik.derivatives(uvi_train, y_train, np.asarray(f_hat, dtype=float).reshape(nx, 1))
    mean = Ks.T @ np.linalg.solve(Ke, np.asarray(f_hat, dtype=float).reshape(nx))
    T = np.linalg.solve(np.eye(nx) + W @ Ke, W @ Ks)
    var = variance - np.einsum("ij,ij->j", Ks, T)
    return mean, var


def demo_problem(n_train=20, true_sigma=0.05, seed=1, d=1):
    """Inputs exactly as GP_preference_demo.py:7-44 for d=1; x@1/d sine for d>1."""
    np.random.seed(seed)

    def true_function(x):
        return np.sin(x * 2 * np.pi + np.pi / 4) + 0.2

    x_train = np.random.random((2 * n_train, d))
    uvi_train = np.random.choice(range(2 * n_train), (n_train, 2), replace=False)
    s = x_train.mean(axis=1)  # x @ (1/d)
    uv = s[uvi_train]
    fuv = true_function(uv) + np.random.normal(scale=true_sigma, size=uv.shape)
    y = -1 * np.ones((fuv.shape[0], 1), dtype="int")
    y[fuv[:, 1] > fuv[:, 0]] = 1
    return x_train, uvi_train, y
