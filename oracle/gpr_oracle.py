"""NumPy restatement of the reference regression path (TEST INFRASTRUCTURE ONLY).

Restates, function by function, the arithmetic of the reference's ``GPr.py``
(osurdml/GPtest) so that parity tests can run on a machine where the reference
itself is absent (the GPU box).  Pinned against the real ``GPr`` import by
``tests/golden/make_golden.py`` -> ``tests/golden/gpr_*.npz`` (see
``tests/test_oracle_golden.py``).

Two flavours are provided:

* ``*_verbatim``  -- the same LAPACK call sequence as the reference
  (``inv`` for prediction GPr.py:48, ``cholesky`` + two LU ``solve`` against
  ``eye(n)`` for the likelihood GPr.py:62-63, ``tile`` based distance
  GPr.py:4-13).  This is the *timed CPU baseline* and the parity oracle.
* ``*_efficient`` -- Cholesky + triangular solves (the algorithm the HIP
  path implements), used to show the >=10x target against an honest CPU
  implementation too.

d > 1 extension: the reference's ``squared_distance`` reshapes to (len,1)
(GPr.py:5-6) and is therefore 1-D only.  For (N,d) inputs the oracle sums the
reference's own per-dimension arithmetic over the d columns;
``x / M`` already broadcasts for (N,d)/(d,) (GPr.py:100,106-107).
"""
import numpy as np


def _sqdist_1d(A, B):
    # GPr.py:4-13, line for line in meaning: A2 + B2 - 2AB with tile temporaries.
    A = np.reshape(A, (len(A), 1))
    B = np.reshape(B, (len(B), 1))
    A2_sum = A * A
    B2_sum = B * B
    AB = 2 * np.dot(A, B.T)
    A2 = np.tile(A2_sum, (1, np.size(B2_sum, axis=0)))
    B2 = np.tile(B2_sum.T, (np.size(A2_sum, axis=0), 1))
    return A2 + B2 - AB


def squared_distance(A, B):
    """GPr.py:4-13 for 1-D inputs; sum over columns of the same arithmetic for (N,d)."""
    A = np.asarray(A, dtype=float)
    B = np.asarray(B, dtype=float)
    if A.ndim == 2 and A.shape[1] > 1:
        out = _sqdist_1d(A[:, 0], B[:, 0])
        for k in range(1, A.shape[1]):
            out = out + _sqdist_1d(A[:, k], B[:, k])
        return out
    return _sqdist_1d(A, B)


def decode_hyp(log_hyp):
    """GPr.py:90-97: hyp = exp(logHyp); M = hyp[:n-2]; sf2 = hyp[n-2]**2; sn2 = hyp[n-1]**2."""
    hyp = np.exp(np.asarray(log_hyp, dtype=float))
    n = len(hyp)
    return hyp[: n - 2], hyp[n - 2] ** 2, hyp[n - 1] ** 2


def kxx(log_hyp, x):
    """GPr.py:99-103."""
    M, sf2, sn2 = decode_hyp(log_hyp)
    scaledX = x / M
    sqDist = squared_distance(scaledX, scaledX)
    return sn2 * np.eye(np.size(x, axis=0)) + sf2 * np.exp(-0.5 * sqDist)


def kxz(log_hyp, x, z):
    """GPr.py:105-110."""
    M, sf2, _ = decode_hyp(log_hyp)
    sqDist = squared_distance(x / M, z / M)
    return sf2 * np.exp(-0.5 * sqDist)


def compute_prediction_verbatim(log_hyp, x, y, z):
    """GPr.py:45-54 (explicit inverse; latent variance, no +sn2)."""
    _, sf2, _ = decode_hyp(log_hyp)
    Kxz = kxz(log_hyp, x, z)
    Kxx = kxx(log_hyp, x)
    iKxx = np.linalg.inv(Kxx)
    Kzx = Kxz.T
    K_diag = np.diagonal(np.dot(np.dot(Kzx, iKxx), Kxz))
    K_noise = sf2 * np.ones(np.size(z, axis=0))
    fz = np.dot(np.dot(Kzx, iKxx), np.asarray(y).T)
    return fz, K_noise - K_diag


def compute_likelihood_verbatim(log_hyp, x, y):
    """GPr.py:57-69; returns the (1,1) ndarray the reference returns."""
    n = np.size(x, axis=0)
    Kxx = kxx(log_hyp, x)
    m = 0.0  # MeanFunction(0).y, GPr.py:72-75
    L = np.linalg.cholesky(Kxx)
    iKxx = np.linalg.solve(L.T, np.linalg.solve(L, np.eye(n)))
    yy = np.reshape(y, (len(y), 1))
    err_y = np.dot(np.dot((yy - m).T, iKxx), (yy - m)) / 2
    det_Kxx = np.sum(np.log(np.diag(L)))
    occams_razor = n * np.log(2 * np.pi) / 2
    return err_y + det_Kxx + occams_razor


def _k_direct(log_hyp, x, z=None):
    M, sf2, sn2 = decode_hyp(log_hyp)
    xs = np.asarray(x, dtype=float).reshape(len(x), -1) / M
    zs = xs if z is None else np.asarray(z, dtype=float).reshape(len(z), -1) / M
    d2 = np.zeros((xs.shape[0], zs.shape[0]))
    for k in range(xs.shape[1]):
        diff = xs[:, k : k + 1] - zs[:, k : k + 1].T
        d2 += diff * diff
    K = sf2 * np.exp(-0.5 * d2)
    if z is None:
        K[np.diag_indices_from(K)] += sn2
    return K


def posterior_efficient(log_hyp, x, y, z):
    """One K build + one Cholesky shared by nlml, mean and variance.

    Same mathematics as GPr.py:45-69, organised the way the HIP path is:
    L = chol(K); w = L^-1 y; nlml = w.w/2 + sum log diag L + n log(2pi)/2;
    V = L^-1 Kxz; fz = V^T w; cov = sf2 - colsum(V*V).
    """
    from scipy.linalg import cholesky, solve_triangular

    _, sf2, _ = decode_hyp(log_hyp)
    n = len(x)
    K = _k_direct(log_hyp, x)
    L = cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    w = solve_triangular(L, np.asarray(y, dtype=float).reshape(n), lower=True, check_finite=False)
    nlml = 0.5 * float(w @ w) + float(np.sum(np.log(np.diag(L)))) + n * np.log(2 * np.pi) / 2
    Kxz = _k_direct(log_hyp, x, z)
    V = solve_triangular(L, Kxz, lower=True, overwrite_b=True, check_finite=False)
    fz = V.T @ w
    cov = sf2 - np.einsum("ij,ij->j", V, V)
    return nlml, fz, cov


def posterior_efficient_blocked(log_hyp, x, y, z, nb=4096):
    """posterior_efficient with the factorisation and the substitutions written over nb-wide blocks (LAPACK only ever
    sees blocks): the flavour used for the N = 32768 fixture, where one monolithic SciPy Cholesky call crashes inside
    OpenBLAS in this image.  Same mathematics (GPr.py:45-69); checked against posterior_efficient in
    tests/test_oracle_golden.py."""
    from scipy.linalg import cholesky, solve_triangular

    _, sf2, _ = decode_hyp(log_hyp)
    n = len(x)
    A = _k_direct(log_hyp, x)                      # becomes L in place (lower blocks)
    starts = list(range(0, n, nb))
    for k0 in starts:
        k1 = min(n, k0 + nb)
        Lkk = cholesky(A[k0:k1, k0:k1], lower=True, check_finite=False)
        A[k0:k1, k0:k1] = Lkk
        if k1 < n:
            A[k1:, k0:k1] = solve_triangular(Lkk, A[k1:, k0:k1].T, lower=True, check_finite=False).T
            for j0 in starts:
                if j0 < k1:
                    continue
                j1 = min(n, j0 + nb)
                A[j0:, j0:j1] -= A[j0:, k0:k1] @ A[j0:j1, k0:k1].T

    def forward(B):                                # L^-1 B, block forward substitution
        B = np.array(B, dtype=float, copy=True)
        for k0 in starts:
            k1 = min(n, k0 + nb)
            B[k0:k1] = solve_triangular(A[k0:k1, k0:k1], B[k0:k1], lower=True, check_finite=False)
            if k1 < n:
                B[k1:] -= A[k1:, k0:k1] @ B[k0:k1]
        return B

    w = forward(np.asarray(y, dtype=float).reshape(n))
    logdet = float(np.sum(np.log(np.diag(A))))
    nlml = 0.5 * float(w @ w) + logdet + n * np.log(2 * np.pi) / 2
    V = forward(_k_direct(log_hyp, x, z))
    fz = V.T @ w
    cov = sf2 - np.einsum("ij,ij->j", V, V)
    return nlml, fz, cov


def nlml_gradient(log_hyp, x, y):
    """d nlml / d log_hyp of GPr.py:57-69 (analytic; the reference has no gradient, so this row is pinned by
    central differences of compute_likelihood_verbatim in tests/test_oracle_golden.py, not by a fixture).

    d nlml / d t = 1/2 tr((Kinv - a a^T) dK/dt), a = Kinv y, with, for Kf = sf2 exp(-r2/2) (GPr.py:102):
    dK/dlog l_k = Kf * (x_ik - x_jk)^2 / l_k^2,  dK/dlog sf = 2 Kf,  dK/dlog sn = 2 sn2 I.
    """
    M, sf2, sn2 = decode_hyp(log_hyp)
    xs = np.asarray(x, dtype=float).reshape(len(x), -1)
    n, d = xs.shape
    K = _k_direct(log_hyp, xs)
    Kf = K - sn2 * np.eye(n)
    Kinv = np.linalg.inv(K)
    a = Kinv @ np.asarray(y, dtype=float).reshape(n)
    A = Kinv - np.outer(a, a)
    g = np.zeros(d + 2)
    for k in range(d):
        diff = (xs[:, k : k + 1] - xs[:, k : k + 1].T) / M[k]
        g[k] = 0.5 * np.sum(A * Kf * diff * diff)
    g[d] = np.sum(A * Kf)
    g[d + 1] = sn2 * np.trace(A)
    return g


def posterior_verbatim(log_hyp, x, y, z):
    """The reference's cost: one compute_likelihood + one compute_prediction (no sharing)."""
    nlml = compute_likelihood_verbatim(log_hyp, x, y)
    fz, cov = compute_prediction_verbatim(log_hyp, x, y, z)
    return float(nlml.flat[0]), fz, cov


def synthetic_problem(N, d, M=1024, seed=1234):
    """SURVEY.md 8(d) / BASELINE.md 3 synthetic inputs (deterministic)."""
    rng = np.random.RandomState(seed)
    X = rng.random_sample((N, d))
    w = rng.randn(d)
    y = np.sin(X @ w) + 0.1 * rng.randn(N)
    Z = rng.random_sample((M, d))
    log_hyp = np.log([0.7] * d + [1.0, 0.1])
    return X, y, Z, log_hyp


def gpy_regression_predict(X, Y, variance, lengthscale, noise_var, Xnew):
    """Dense restatement of the GPy calls of GP_parameter_fit.py:30-31,52 (GPy is absent: PARITY UNPINNED; GPy's
    documented semantics): RBF kernel variance*exp(-r^2/2), Gaussian noise; returns (mean (M,1), var of y (M,1),
    log marginal likelihood).  Same SE arithmetic as GPr.py:99-110 with sf2 = variance, sn2 = noise_var."""
    X = np.asarray(X, dtype=float).reshape(len(X), -1)
    d = X.shape[1]
    ell = np.asarray(lengthscale, dtype=float).reshape(-1) * np.ones(d)
    lh = np.concatenate([np.log(ell), [0.5 * np.log(variance)], [0.5 * np.log(noise_var)]])
    nlml, fz, cov = posterior_efficient(lh, X, np.asarray(Y, dtype=float).reshape(-1), np.asarray(Xnew, dtype=float).reshape(len(Xnew), -1))
    return fz.reshape(-1, 1), (cov + noise_var).reshape(-1, 1), -nlml
