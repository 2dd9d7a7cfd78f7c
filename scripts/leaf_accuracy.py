import ctypes, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from gptest_amd import _lib
from gpk_hooks import hooks
_lib.init()
P = _lib.c_double_p
ptr = lambda a: a.ctypes.data_as(P)
rng = np.random.RandomState(0)
worst = 0.0; worst_inv = 0.0
for trial in range(20):
    d = 10.0 ** rng.uniform(-8, 8, 128)
    A = np.diag(d)
    L = np.zeros((128, 128)); Di = np.zeros((128, 128))
    rc = hooks().gpk_test_potf2_128(ptr(A), ptr(L), ptr(Di), trial & 1)
    assert rc == 0
    ref = np.sqrt(d)
    ulp = np.abs(np.diag(L) - ref) / np.spacing(ref)
    worst = max(worst, ulp.max())
    refi = 1.0 / ref
    ulpi = np.abs(np.diag(Di) - refi) / np.spacing(refi)
    worst_inv = max(worst_inv, ulpi.max())
print("diagonal tiles: sqrt error max %.2f ulp, 1/sqrt (Dinv diagonal) max %.2f ulp" % (worst, worst_inv))
# dense SPD tile: residual
Q = rng.randn(128, 128); A = Q @ Q.T + 128 * np.eye(128)
L = np.zeros((128, 128)); Di = np.zeros((128, 128))
hooks().gpk_test_potf2_128(ptr(A), ptr(L), ptr(Di), 0)
Lr = np.linalg.cholesky(A)
print("dense tile: max |L - L_lapack| / max|L| = %.2e, ||L L^T - A|| / ||A|| = %.2e (lapack: %.2e), ||Dinv L - I|| = %.2e" % (
    np.abs(np.tril(L) - Lr).max() / np.abs(Lr).max(), np.linalg.norm(np.tril(L) @ np.tril(L).T - A) / np.linalg.norm(A),
    np.linalg.norm(Lr @ Lr.T - A) / np.linalg.norm(A), np.linalg.norm(np.tril(Di) @ np.tril(L) - np.eye(128))))
