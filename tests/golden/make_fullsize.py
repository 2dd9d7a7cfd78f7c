"""Full-size fixtures for BASELINE configs 2, 3 (and 4's size) (container-only script; c2 + c3: ~2 min, ~6 GiB of RAM).

    python tests/golden/make_fullsize.py            # c2 c3
    python tests/golden/make_fullsize.py c4         # N = 32768: ~10 min, ~30 GiB

Writes ``full_c3_16384_d8.npz`` (and ``full_c2_4096_d4.npz``): nlml, posterior mean and latent variance of the
SURVEY 8(d) synthetic problem at M = 1024 test points, computed by ``oracle.gpr_oracle.posterior_efficient``
(SciPy Cholesky + triangular solves on the CPU).

Provenance, stated plainly: these numbers are RESTATEMENT-DERIVED at a size the reference itself cannot reach in
reasonable time (its explicit inverses at N = 16384 take ~15 min and 3x the memory).  The efficient flavour is
pinned to the reference at N <= 512 by tests/test_oracle_golden.py::test_efficient_oracle_matches_reference
against fixtures written by the real ``GPr.py`` import (make_golden.py); fp64 Cholesky of this K (cond ~1e5)
reproduces nlml to ~1e-13 relative whatever the summation order.  Inputs are NOT stored: they are regenerated
from the seed by ``synthetic_problem`` (checked through the stored checksums).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import gpr_oracle as orc  # noqa: E402


def make(name, N, d, M, blocked=False):
    X, y, Z, lh = orc.synthetic_problem(N, d, M)
    t0 = time.perf_counter()
    nlml, fz, cov = (orc.posterior_efficient_blocked if blocked else orc.posterior_efficient)(lh, X, y, Z)
    dt = time.perf_counter() - t0
    path = os.path.join(HERE, "full_%s.npz" % name)
    np.savez(path, N=N, d=d, M=M, seed=1234, log_hyp=lh, nlml=nlml, fz=fz, cov_fz=cov,
             checksum_X=float(X.sum()), checksum_y=float(y.sum()), checksum_Z=float(Z.sum()),
             provenance="oracle.gpr_oracle.posterior_efficient%s (restatement-derived; flavour pinned to the real "
                        "GPr.py at N<=512 by test_efficient_oracle_matches_reference)" % ("_blocked" if blocked else ""))
    print("%s: nlml = %.13f  (%.1f s) -> %s" % (name, nlml, dt, path))


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c3"]
    if "c2" in which:
        make("c2_4096_d4", 4096, 4, 1024)
    if "c3" in which:
        make("c3_16384_d8", 16384, 8, 1024)
    if "c4" in which:   # BASELINE config 4's size: ~10 min and ~30 GiB of RAM on 8 cores
        make("c4_32768_d8", 32768, 8, 1024, blocked=True)   # one monolithic SciPy Cholesky of this size crashes in OpenBLAS here
