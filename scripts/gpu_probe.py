"""First-contact diagnostics on the GPU box: run every kernel once against NumPy and print
max errors (no asserts, so one gpurun shows the state of everything)."""
import os
import sys
import time
import ctypes

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib  # noqa: E402
from gptest_amd import GPr  # noqa: E402
from oracle import gpr_oracle as orc  # noqa: E402

P = _lib.c_double_p


def ptr(a):
    return a.ctypes.data_as(P)


def main():
    _lib.init()
    l = _lib.lib()
    rng = np.random.RandomState(0)

    # --- MFMA GEMM
    for (m, n, k, tile, sub, lower) in [(128, 128, 16, 128, 0, 0), (128, 128, 64, 128, 1, 0), (64, 64, 16, 64, 0, 0),
                                        (256, 128, 128, 64, 1, 0), (384, 384, 256, 128, 1, 1), (256, 256, 32, 64, 1, 1)]:
        A = rng.randn(m, k)
        B = rng.randn(n, k) if not lower else A.copy()
        C0 = rng.randn(m, n)
        C = C0.copy()
        rc = l.gpk_test_gemm_nt(ptr(A), ptr(B), ptr(C), m, n, k, sub, lower, tile)
        ref = (C0 - A @ B.T) if sub else A @ B.T
        if lower:
            bt = tile
            mask = np.zeros((m, n), bool)
            for ti in range(m // bt):
                for tj in range(ti + 1):
                    mask[ti * bt:(ti + 1) * bt, tj * bt:(tj + 1) * bt] = True
            err = np.abs(C - ref)[mask].max()
            untouched = np.abs(C - C0)[~mask].max() if (~mask).any() else 0.0
            print("gemm m=%d n=%d k=%d tile=%d sub=%d lower: rc=%d err=%.3e untouched=%.3e" % (m, n, k, tile, sub, rc, err, untouched))
        else:
            print("gemm m=%d n=%d k=%d tile=%d sub=%d: rc=%d err=%.3e" % (m, n, k, tile, sub, rc, np.abs(C - ref).max()))
            if np.abs(C - ref).max() > 1e-9 and m <= 128:
                # layout debugging aid: which permutation?
                print("   C[0,:4]", C[0, :4], "ref", ref[0, :4], "refT", ref.T[0, :4])

    # --- leaf factor + inverse
    Q = rng.randn(128, 128)
    A = Q @ Q.T + 128 * np.eye(128)
    L = np.zeros((128, 128))
    D = np.zeros((128, 128))
    rc = l.gpk_test_potf2_128(ptr(A), ptr(L), ptr(D))
    Lr = np.linalg.cholesky(A)
    print("potf2: rc=%d |L-Lref|=%.3e |Dinv*L-I|=%.3e upper(L)=%.1e upper(D)=%.1e" % (
        rc, np.abs(L - Lr).max(), np.abs(D @ Lr - np.eye(128)).max(), np.abs(np.triu(L, 1)).max(), np.abs(np.triu(D, 1)).max()))
    Abad = A.copy()
    Abad[5, 5] = -1.0
    rc = l.gpk_test_potf2_128(ptr(Abad), ptr(L), ptr(D))
    print("potf2 not-PD: rc=%d (expect 6)" % rc)

    # --- builders
    for (N, M, d) in [(20, 7, 1), (200, 50, 3), (300, 129, 8), (130, 5, 11)]:
        X = rng.rand(N, d)
        Z = rng.rand(M, d)
        lh = np.log(np.r_[0.3 + rng.rand(d), 1.3, 0.2])
        cov = GPr.SquaredExponential(lh, X)
        K = cov.compute_Kxx_matrix()
        Kz = cov.compute_Kxz_matrix(Z)
        D2 = GPr.squared_distance(X, Z)
        print("build N=%d M=%d d=%d: Kxx %.2e Kxz %.2e sqdist %.2e" % (
            N, M, d, np.abs(K - orc.kxx(lh, X)).max(), np.abs(Kz - orc.kxz(lh, X, Z)).max(),
            np.abs(D2 - orc.squared_distance(X, Z)).max()))

    # --- potrf through the public ABI
    for N in (1, 5, 128, 200, 384, 640, 1000):
        Q = rng.randn(N, N)
        A = Q @ Q.T + N * np.eye(N)
        Ain = A.copy()
        rc = l.gpk_potrf(ptr(Ain), N)
        Lr = np.linalg.cholesky(A)
        print("potrf N=%d rc=%d rel err %.3e" % (N, rc, np.abs(Ain - Lr).max() / np.abs(Lr).max()))

    # --- fit / predict vs oracle
    for (N, d, M) in [(20, 1, 100), (300, 3, 50), (1000, 4, 200), (2048, 8, 256)]:
        X, y, Z, lh = orc.synthetic_problem(N, d, M)
        gp = GPr.GaussianProcess(lh, 0, 0, "SE", "zero", "zero", X, y)
        t0 = time.time()
        nl = gp.compute_likelihood(lh)
        fz, cov = gp.compute_prediction(Z)
        t1 = time.time()
        nl_o, fz_o, cov_o = orc.posterior_efficient(lh, X, y, Z)
        print("fit N=%d d=%d: nlml %.12g vs %.12g (rel %.2e) mean %.2e var %.2e  [%.1f ms]" % (
            N, d, nl[0, 0], nl_o, abs(nl[0, 0] - nl_o) / abs(nl_o), np.abs(fz - fz_o).max() / np.abs(fz_o).max(),
            np.abs(cov - cov_o).max(), 1e3 * (t1 - t0)))

    # --- timing at C2 / C3
    l.gpk_set_option(b"timing", 1.0)
    for (N, d, M) in [(4096, 4, 1024), (16384, 8, 1024)]:
        X, y, Z, lh = orc.synthetic_problem(N, d, M)
        ms = ctypes.c_double()
        nl = ctypes.c_double()
        mean = np.zeros(M)
        var = np.zeros(M)
        Xc, yc, Zc, lhc = (np.ascontiguousarray(a) for a in (X, y, Z, lh))
        rc = l.gpk_bench_posterior(ptr(Xc), ptr(yc), N, d, ptr(lhc), ptr(Zc), M, 1, 3, ctypes.byref(ms), ctypes.byref(nl), ptr(mean), ptr(var))
        print("bench N=%d d=%d rc=%d: %.2f ms/posterior nlml=%.10g" % (N, d, rc, ms.value / 3, nl.value), _lib.last_error() if rc else "")
        for st in ("kbuild", "potrf", "panel", "trailing", "solve_y", "predict"):
            a = ctypes.c_double()
            c = ctypes.c_double()
            l.gpk_last_timing(st.encode(), ctypes.byref(a), ctypes.byref(c))
            print("   %-9s %.3f ms total over %d launches" % (st, a.value / 3, int(c.value)))
        if N <= 4096:
            nl_o, fz_o, cov_o = orc.posterior_efficient(lh, X, y, Z)
            print("   parity: nlml rel %.2e mean %.2e var %.2e" % (abs(nl.value - nl_o) / abs(nl_o), np.abs(mean - fz_o).max() / np.abs(fz_o).max(), np.abs(var - cov_o).max()))
        flops = sum((N - (k + 1) * 256) ** 2 * 256 for k in range(N // 256))  # lower+upper counted once each way
        a = ctypes.c_double(); c = ctypes.c_double()
        l.gpk_last_timing(b"trailing", ctypes.byref(a), ctypes.byref(c))
        if a.value > 0:
            print("   trailing update: %.2f TFLOP/s (algorithmic lower-triangle flops)" % (flops / (a.value / 3 * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
