"""Generate golden vectors from the REAL reference (container-only script).

Imports ``/root/reference/GPr.py`` (pure NumPy) from its local path and writes
small ``.npz`` fixtures holding inputs and the reference's own outputs.  The
reference cannot travel to the GPU box, the fixtures can.  Run:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

d > 1 cases: ``GPr.squared_distance`` is 1-D only (GPr.py:5-6).  The module
global is rebound to "sum of the reference's own 1-D function over columns"
and the *unmodified* ``GaussianProcess.compute_prediction`` /
``compute_likelihood`` are called (they look the function up at call time,
GPr.py:101,108).  Every arithmetic step is still the reference's.

``GPpref.py`` is Python-2 + GPy and cannot be imported; its
fixtures are produced by ``oracle/gppref_oracle.py`` and are labelled
"restatement-derived" -- parity for that path is unpinned.
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sys.path.insert(0, REF)
    import GPr  # noqa: E402  (the real reference module)

    assert os.path.dirname(os.path.abspath(GPr.__file__)) == REF
    orig = GPr.squared_distance

    def sqdist_nd(A, B):
        A = np.asarray(A)
        B = np.asarray(B)
        if A.ndim == 2 and A.shape[1] > 1:
            out = orig(A[:, 0], B[:, 0])
            for k in range(1, A.shape[1]):
                out = out + orig(A[:, k], B[:, k])
            return out
        return orig(A, B)

    GPr.squared_distance = sqdist_nd
    return GPr


def run_case(GPr, name, x, y, z, log_hyp, keep_k=True):
    gp = GPr.GaussianProcess(log_hyp, 0, 0, "SE", "zero", "zero", x, y)
    nlml = gp.compute_likelihood(log_hyp)
    fz, cov = gp.compute_prediction(z)
    out = dict(x=x, y=y, z=z, log_hyp=np.asarray(log_hyp, float), nlml=np.asarray(nlml),
               fz=fz, cov_fz=cov, M=gp.covFun.M, sf2=gp.covFun.sf2, sn2=gp.covFun.sn2)
    if keep_k:
        out["Kxx"] = gp.covFun.compute_Kxx_matrix()
        out["Kxz"] = gp.covFun.compute_Kxz_matrix(z)
        out["sqdist_xz"] = GPr.squared_distance(x / gp.covFun.M, z / gp.covFun.M)
    path = os.path.join(HERE, "gpr_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-22s N=%-4d nlml=%.15g fz0=%.15g cov0=%.15g" % (name, len(x), float(np.asarray(nlml).flat[0]), fz[0], cov[0]))


def main():
    GPr = load_reference()
    c = np.array([3, 5, -9, -3, 2], float)

    # (1) the seeded demo, exactly GP_regression_demo.py:6-36
    np.random.seed(0)
    x = np.random.random(20)
    y = np.polyval(c, x) + np.random.normal(0, 0.1, len(x))
    z = np.arange(0, 1, 0.01, float)
    run_case(GPr, "demo20_d1", x, y, z, np.log([1, 1, 0.1]))

    # (2) BASELINE config 1: N=512, d=1, demo polynomial target
    rng = np.random.RandomState(0)
    x = rng.random_sample(512)
    y = np.polyval(c, x) + rng.normal(0, 0.1, 512)
    run_case(GPr, "c1_512_d1", x, y, z, np.log([1, 1, 0.1]), keep_k=False)

    # (3) (N,1) column-vector input, ragged size (not a multiple of any tile)
    rng = np.random.RandomState(7)
    x = rng.random_sample((77, 1))
    y = np.sin(6 * x[:, 0]) + 0.05 * rng.randn(77)
    z = rng.random_sample((13, 1))
    run_case(GPr, "ragged77_d1col", x, y, z, np.log([0.3, 1.3, 0.2]))

    # (4,5) d>1 cases with the SURVEY 8(d) generator
    for (N, d, M) in ((256, 4, 64), (256, 8, 64), (300, 3, 50)):
        rng = np.random.RandomState(1234)
        X = rng.random_sample((N, d))
        w = rng.randn(d)
        yy = np.sin(X @ w) + 0.1 * rng.randn(N)
        Z = rng.random_sample((M, d))
        run_case(GPr, "syn%d_d%d" % (N, d), X, yy, Z, np.log([0.7] * d + [1.0, 0.1]), keep_k=(N <= 256 and d == 4))

    # (6) ARD with distinct length scales and a poorly conditioned (small-noise) setting
    rng = np.random.RandomState(99)
    X = rng.random_sample((128, 2))
    yy = np.cos(3 * X[:, 0]) * X[:, 1] + 0.01 * rng.randn(128)
    Z = rng.random_sample((32, 2))
    run_case(GPr, "ard128_d2_lownoise", X, yy, Z, np.log([0.4, 1.5, 2.0, 0.02]))

    # (7) tiny edge cases: N=1 and N=2
    run_case(GPr, "tiny1_d1", np.array([0.25]), np.array([0.5]), np.array([0.0, 0.25, 1.0]), np.log([0.5, 1.0, 0.1]))
    run_case(GPr, "tiny2_d1", np.array([0.25, 0.75]), np.array([0.5, -0.5]), np.array([0.1, 0.9]), np.log([0.5, 1.0, 0.1]))


if __name__ == "__main__":
    main()
