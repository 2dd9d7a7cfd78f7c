"""CPU checks of the classification oracle (oracle/gpc_oracle.py).  PARITY UNPINNED (GPc.py does not parse); what can
be pinned without the reference is the mathematics the sketch names: Psi(f) of Rasmussen eq. 3.12 (GPc.py:40-46)."""
import numpy as np
import pytest

from oracle import gpc_oracle as co


@pytest.mark.parametrize("link", ["probit", "logit"])
def test_mode_maximises_psi(link):
    """The Newton loop's fixed point is the maximiser of Psi(f) = log p(y|f) - f^T K^-1 f / 2 (eq. 3.12), found here
    by an independent quasi-Newton search on the whitened problem."""
    from scipy.optimize import minimize

    x, y = co.demo_problem(n_train=25, seed=3, d=1)
    lh = np.log([0.3, 1.2])
    f, lml, it = co.calc_laplace(x, y, lh, link=link, delta_f=1e-12)
    K = co.rbf_ard_K(x, np.exp(lh[:1]), np.exp(lh[1]) ** 2) + 1e-10 * np.eye(25)
    L = np.linalg.cholesky(K)

    def neg_psi(v):          # f = L v  ->  f^T K^-1 f = v.v
        ff = L @ v
        lp, g, _ = co.likelihood_terms(y, ff, link)
        return -(lp.sum() - 0.5 * v @ v), -(L.T @ g - v)

    r = minimize(neg_psi, np.zeros(25), jac=True, method="L-BFGS-B", options=dict(maxiter=2000, ftol=1e-15, gtol=1e-10))
    np.testing.assert_allclose(L @ r.x, f, rtol=0, atol=1e-5 * np.abs(f).max())
    assert it < 20


def test_likelihood_terms_are_derivatives():
    rng = np.random.RandomState(0)
    y = np.where(rng.rand(50) > 0.5, 1.0, -1.0)
    f = 4.0 * rng.randn(50)
    f[:5] = [-40.0, -12.0, 12.0, 40.0, 0.0]          # the erfcx branch and both tails
    h = 1e-5
    for link in ("probit", "logit"):
        lp, g, W = co.likelihood_terms(y, f, link)
        lp1, g1, _ = co.likelihood_terms(y, f + h, link)
        lp0, g0, _ = co.likelihood_terms(y, f - h, link)
        np.testing.assert_allclose(g, (lp1 - lp0) / (2 * h), rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(-W, (g1 - g0) / (2 * h), rtol=1e-5, atol=1e-8)
        assert np.all(W >= 0) and np.all(np.isfinite(lp))


def test_label_preprocessing_follows_the_sketch():
    """GPc.py:24-38."""
    np.testing.assert_array_equal(co.preprocess_labels([0, 1, 1, 0]), [-1, 1, 1, -1])
    np.testing.assert_array_equal(co.preprocess_labels([-1, 1]), [-1, 1])
    with pytest.raises(AssertionError):
        co.preprocess_labels([0, 2])


def test_demo_data_is_the_reference_demo_construction():
    """GP_classification_demo.py:5-41 with np.random.seed(0): first training inputs of the legacy MT19937 stream."""
    x, y = co.demo_problem()
    assert x.shape == (20, 1) and set(np.unique(y)) <= {-1.0, 1.0}
    np.testing.assert_allclose(x[:3, 0], [0.5488135, 0.71518937, 0.60276338], atol=1e-8)


def test_prediction_at_training_points_reproduces_the_mode():
    x, y = co.demo_problem(n_train=60, seed=2, d=2)
    lh = np.log([0.4, 0.4, 1.0])
    f, _, _ = co.calc_laplace(x, y, lh, delta_f=1e-12)
    mean, var, prob = co.predict(x, y, lh, f, x)
    np.testing.assert_allclose(mean, f, rtol=0, atol=1e-9)      # k_i^T grad = (K grad)_i = f_i at the mode
    assert np.all(var > -1e-12) and np.all((prob > 0) & (prob < 1))
