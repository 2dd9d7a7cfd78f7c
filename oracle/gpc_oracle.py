"""NumPy restatement of binary GP classification under the Laplace approximation (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED.  The reference's ``GPc.py`` does not parse (``GPc.py:10`` is a dangling ``y*``, ``GPc.py:16`` writes
``def ClassifierLikelihood(object)``, ``GPc.py:40`` is mis-indented) and ``GP_classification_demo.py`` stops at line 50
with a stray ``l``: there is no runnable reference for this row and no fixture.  What the two files do fix is restated
here with its line numbers: the probit likelihood ``log Phi(y f)`` (GPc.py:4-6), the logistic link
(GPc.py:13-14), the label check / {0,1} -> {-1,+1} map (GPc.py:24-38), "Rasmussen eq. 3.12" as the objective
(GPc.py:40-46) and the demo's seeded data (GP_classification_demo.py:5-41).  The loop itself is Rasmussen & Williams
(2006) Algorithm 3.1, prediction Algorithm 3.2, written densely; the stopping rule is the one of the reference's
sibling Laplace loop (max |f_new - f| <= delta_f, GPpref.py:140).
"""
import numpy as np
from scipy.special import erfcx, ndtr as std_norm_cdf


def logistic_function(z):
    """GPc.py:13-14."""
    return 1.0 / (1 + np.exp(-z))


def preprocess_labels(Y):
    """GPc.py:24-38: labels must be in {0,1} or {-1,1}; 0 becomes -1."""
    Y = np.asarray(Y)
    Y_prep = Y.copy().astype(float)
    Y1 = Y[Y.flatten() == 1].size
    Y2 = Y[Y.flatten() == 0].size
    Y3 = Y[Y.flatten() == -1].size
    assert (Y1 + Y2 == Y.size) or (Y1 + Y3 == Y.size), "Inputs should be in {0,1} or {-1,1}."
    Y_prep[Y.flatten() == 0] = -1
    return Y_prep


def _probit_terms(z):
    """log Phi(z) (GPc.py:5-6) and phi(z)/Phi(z), stable for very negative z."""
    z = np.asarray(z, dtype=float)
    lp = np.empty_like(z)
    r = np.empty_like(z)
    a = z > -5.0
    p = std_norm_cdf(z[a])
    lp[a] = np.log(p)
    r[a] = np.exp(-0.5 * z[a] ** 2) / np.sqrt(2 * np.pi) / p
    t = -z[~a] / np.sqrt(2.0)
    e = erfcx(t)
    lp[~a] = np.log(0.5 * e) - t * t
    r[~a] = np.sqrt(2.0 / np.pi) / e
    return lp, r


def likelihood_terms(y, f, link="probit"):
    """(log p(y|f), d/df, -d2/df2) element-wise, y in {-1,+1}."""
    y = np.asarray(y, dtype=float).ravel()
    f = np.asarray(f, dtype=float).ravel()
    if link == "probit":
        z = y * f
        lp, r = _probit_terms(z)
        return lp, y * r, r * r + z * r
    z = y * f
    lp = np.where(z > 0, -np.log1p(np.exp(-np.abs(z))), z - np.log1p(np.exp(-np.abs(z))))
    pi = logistic_function(f)
    return lp, 0.5 * (y + 1.0) - pi, pi * (1.0 - pi)


def rbf_ard_K(x, lengthscale, variance):
    xs = np.asarray(x, dtype=float).reshape(len(x), -1) / np.asarray(lengthscale, dtype=float)
    d2 = np.zeros((xs.shape[0], xs.shape[0]))
    for k in range(xs.shape[1]):
        diff = xs[:, k:k + 1] - xs[:, k:k + 1].T
        d2 += diff * diff
    return variance * np.exp(-0.5 * d2)


def calc_laplace(x, y, loghyp, link="probit", delta_f=1e-6, max_iter=100, f=None, trace=None):
    """R&W Algorithm 3.1.  loghyp = log([l_1..l_d, sigma_f]).  Returns (f (n,), lml, iterations)."""
    x = np.asarray(x, dtype=float).reshape(len(x), -1)
    y = preprocess_labels(y).ravel()
    n, d = x.shape
    hyp = np.exp(np.asarray(loghyp, dtype=float))
    K = rbf_ard_K(x, hyp[:d], hyp[d] ** 2)
    f = np.zeros(n) if f is None else np.asarray(f, dtype=float).ravel().copy()
    it = 0
    a = np.zeros(n)
    L = np.eye(n)
    while it < max_iter:
        _, g, W = likelihood_terms(y, f, link)
        sW = np.sqrt(W)
        B = np.eye(n) + sW[:, None] * K * sW[None, :]
        L = np.linalg.cholesky(B)
        b = W * f + g
        a = b - sW * np.linalg.solve(L.T, np.linalg.solve(L, sW * (K @ b)))
        f_new = K @ a
        err = np.max(np.abs(f_new - f))
        f = f_new
        it += 1
        if trace is not None:
            trace.append(err)
        if not (err > delta_f):
            break
    lp, _, _ = likelihood_terms(y, f, link)
    lml = -0.5 * float(a @ f) + float(np.sum(lp)) - float(np.sum(np.log(np.diag(L))))   # R&W eq. 3.32 (GPc.py:42-46)
    return f, lml, it


def predict(x, y, loghyp, f_hat, x_test, link="probit"):
    """R&W Algorithm 3.2 at the mode: (latent mean, latent variance, averaged probability of class +1)."""
    x = np.asarray(x, dtype=float).reshape(len(x), -1)
    z = np.asarray(x_test, dtype=float).reshape(len(x_test), -1)
    y = preprocess_labels(y).ravel()
    n, d = x.shape
    hyp = np.exp(np.asarray(loghyp, dtype=float))
    K = rbf_ard_K(x, hyp[:d], hyp[d] ** 2)
    _, g, W = likelihood_terms(y, f_hat, link)
    sW = np.sqrt(W)
    L = np.linalg.cholesky(np.eye(n) + sW[:, None] * K * sW[None, :])
    xs, zs = x / hyp[:d], z / hyp[:d]
    d2 = np.zeros((len(z), n))
    for k in range(d):
        diff = zs[:, k:k + 1] - xs[:, k:k + 1].T
        d2 += diff * diff
    Ks = hyp[d] ** 2 * np.exp(-0.5 * d2)          # (M, n)
    mean = Ks @ g
    V = np.linalg.solve(L, (Ks * sW[None, :]).T)  # (n, M)
    var = hyp[d] ** 2 - np.sum(V * V, axis=0)
    if link == "probit":
        prob = std_norm_cdf(mean / np.sqrt(1.0 + np.maximum(var, 0.0)))
    else:
        prob = logistic_function(mean / np.sqrt(1.0 + np.pi * np.maximum(var, 0.0) / 8.0))
    return mean, var, prob


def demo_problem(n_train=20, seed=0, d=1):
    """Training data exactly as GP_classification_demo.py:5-41 for d = 1 (np.random.seed(0), 20 uniform inputs, labels
    drawn with probability Phi(true_function(x))); the mean over columns feeds true_function for d > 1."""
    np.random.seed(seed)

    def true_function(x):
        return np.sin(x * 2 * np.pi + np.pi / 4) + 0.2

    if d == 1:
        x_train = np.random.random(n_train)
        fx = std_norm_cdf(true_function(x_train))
        y_train = np.array([np.random.uniform() < p for p in fx])
        return x_train.reshape(-1, 1), np.where(y_train, 1.0, -1.0)
    x_train = np.random.random((n_train, d))
    fx = std_norm_cdf(true_function(x_train.mean(axis=1)))
    y_train = np.random.uniform(size=n_train) < fx
    return x_train, np.where(y_train, 1.0, -1.0)
