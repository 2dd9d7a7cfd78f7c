"""CPU: the oracle (NumPy restatement) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py imports /root/reference/GPr.py).  This pins the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_cases
from oracle import gpr_oracle as orc


@pytest.mark.parametrize("case", golden_cases())
def test_verbatim_oracle_matches_reference(case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    x, y, z, lh = g["x"], g["y"], g["z"], g["log_hyp"]
    nlml = orc.compute_likelihood_verbatim(lh, x, y)
    assert nlml.shape == (1, 1)  # GPr.py:65-69 returns a (1,1) array
    # identical LAPACK call sequence on the same machine -> bit-for-bit or last-ulp
    np.testing.assert_allclose(nlml, g["nlml"], rtol=1e-13, atol=0)
    fz, cov = orc.compute_prediction_verbatim(lh, x, y, z)
    np.testing.assert_allclose(fz, g["fz"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(cov, g["cov_fz"], rtol=0, atol=1e-12 * float(g["sf2"]))
    M, sf2, sn2 = orc.decode_hyp(lh)
    np.testing.assert_array_equal(M, g["M"])
    assert sf2 == g["sf2"] and sn2 == g["sn2"]
    if "Kxx" in g.files:
        np.testing.assert_array_equal(orc.kxx(lh, x), g["Kxx"])
        np.testing.assert_array_equal(orc.kxz(lh, x, z), g["Kxz"])
        np.testing.assert_array_equal(orc.squared_distance(x / M, z / M), g["sqdist_xz"])


@pytest.mark.parametrize("case", golden_cases())
def test_efficient_oracle_matches_reference(case):
    """The Cholesky/triangular-solve organisation (what the HIP path computes) against the
    reference's inv/LU organisation: tolerance 1e-6 relative (BASELINE.json north_star);
    variance tolerance relative to sf2 because of cancellation near training points."""
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    nlml, fz, cov = orc.posterior_efficient(g["log_hyp"], g["x"], g["y"], g["z"])
    assert abs(nlml - float(g["nlml"].flat[0])) <= 1e-6 * max(1.0, abs(float(g["nlml"].flat[0])))
    np.testing.assert_allclose(fz, g["fz"], rtol=1e-6, atol=1e-6 * np.abs(g["fz"]).max())
    np.testing.assert_allclose(cov, g["cov_fz"], rtol=0, atol=1e-6 * float(g["sf2"]))


def test_survey_known_answers():
    """KATs recorded in SURVEY.md section 4 from the real import (seeded demo)."""
    g = np.load(os.path.join(GOLDEN, "gpr_demo20_d1.npz"))
    assert float(g["nlml"].flat[0]) == pytest.approx(40.891674532620144, rel=1e-14)
    assert g["fz"][0] == pytest.approx(2.100415843582149, rel=1e-13)
    assert g["fz"][50] == pytest.approx(-0.9930107642771688, rel=1e-13)
    assert g["cov_fz"][99] == pytest.approx(0.003276654316368477, rel=1e-10)
    assert g["Kxx"][0, 0] == 1.01 and g["Kxx"][3, 7] == pytest.approx(0.9416079567368075, rel=1e-15)
    c1 = np.load(os.path.join(GOLDEN, "gpr_c1_512_d1.npz"))
    assert float(c1["nlml"].flat[0]) == pytest.approx(-226.50248887770795, rel=1e-13)


def test_synthetic_problem_is_deterministic():
    X, y, Z, lh = orc.synthetic_problem(64, 8, 16)
    X2, y2, Z2, lh2 = orc.synthetic_problem(64, 8, 16)
    assert np.array_equal(X, X2) and np.array_equal(y, y2) and np.array_equal(Z, Z2)
    assert X.shape == (64, 8) and Z.shape == (16, 8) and len(lh) == 10
    g = np.load(os.path.join(GOLDEN, "gpr_syn256_d8.npz"))
    Xg, yg, _, _ = orc.synthetic_problem(256, 8, 64)
    assert np.array_equal(Xg, g["x"]) and np.array_equal(yg, g["y"])


def test_gppref_oracle_seeded_demo_trace():
    """PARITY UNPINNED (GPpref.py is Python-2 + GPy): self-consistency with the trace recorded in
    SURVEY.md 8(c) from an independent transcription -- restatement-derived, not reference-derived."""
    from oracle import gppref_oracle as po

    x, uvi, y = po.demo_problem()
    tr = []
    f, lml, it, eps = po.calc_laplace(x, uvi, y, np.log([0.1, 1.0, 0.1]), delta_f=1e-5, trace=tr)
    assert it == 5 and eps == 1e-6
    assert lml == pytest.approx(40.5831069552, rel=1e-9)
    np.testing.assert_allclose(f[:5, 0], [-0.0761991, -1.31395028, 0.59264976, 0.47781206, 1.24381158], atol=5e-8)
    np.testing.assert_allclose([t[0] for t in tr[:4]], [1.137, 0.2108, 8.14e-3, 1.20e-5], rtol=2e-3)


@pytest.mark.parametrize("N,d,seed", [(40, 1, 0), (60, 3, 1), (90, 8, 2)])
def test_oracle_gradient_matches_central_differences(N, d, seed):
    """The reference has no gradient; the analytic oracle is pinned on the reference-restated nlml
    (compute_likelihood_verbatim, itself pinned by the golden files) through central differences."""
    x, y, _, lh = orc.synthetic_problem(N, d, M=4, seed=seed)
    g = orc.nlml_gradient(lh, x, y)
    h = 1e-5
    for k in range(d + 2):
        e = np.zeros(d + 2)
        e[k] = h
        fd = (orc.compute_likelihood_verbatim(lh + e, x, y).flat[0]
              - orc.compute_likelihood_verbatim(lh - e, x, y).flat[0]) / (2 * h)
        assert abs(g[k] - fd) <= 1e-6 * max(1.0, abs(fd))


def test_blocked_efficient_flavour_matches_the_monolithic_one():
    """posterior_efficient_blocked (used for the N = 32768 fixture, where one monolithic SciPy Cholesky crashes inside
    OpenBLAS in this image) against posterior_efficient on a ragged block split, and against the committed C2 fixture."""
    import os

    from conftest import GOLDEN

    X, y, Z, lh = orc.synthetic_problem(1300, 3, 40)
    a = orc.posterior_efficient(lh, X, y, Z)
    b = orc.posterior_efficient_blocked(lh, X, y, Z, nb=300)
    assert abs(a[0] - b[0]) <= 1e-12 * abs(a[0])
    np.testing.assert_allclose(b[1], a[1], rtol=0, atol=1e-11 * np.abs(a[1]).max())
    np.testing.assert_allclose(b[2], a[2], rtol=0, atol=1e-12)
    g = np.load(os.path.join(GOLDEN, "full_c2_4096_d4.npz"))
    X, y, Z, lh = orc.synthetic_problem(4096, 4, 1024)
    c = orc.posterior_efficient_blocked(lh, X, y, Z, nb=1024)
    assert abs(c[0] - float(g["nlml"])) <= 1e-12 * abs(float(g["nlml"]))
    np.testing.assert_allclose(c[1], g["fz"], rtol=0, atol=1e-10 * np.abs(g["fz"]).max())
    np.testing.assert_allclose(c[2], g["cov_fz"], rtol=0, atol=1e-11)
