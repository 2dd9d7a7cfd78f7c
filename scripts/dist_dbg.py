import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
P_ = _lib.c_double_p
for (N, P, nb) in [(2048, 4, 256), (2048, 2, 256), (2048, 8, 256), (1792, 4, 256), (2304, 4, 256), (2500, 8, 128), (3000, 8, 256)]:
    rng = np.random.RandomState(N + P)
    Q = rng.randn(N, N); A = Q @ Q.T + N * np.eye(N)
    _lib.set_option("nb", nb)
    L1 = np.empty((N, N)); LP = np.empty((N, N))
    rc1 = l.gpk_test_dist_potrf(A.ctypes.data_as(P_), N, 1, L1.ctypes.data_as(P_))
    rcp = l.gpk_test_dist_potrf(A.ctypes.data_as(P_), N, P, LP.ctypes.data_as(P_))
    Lr = np.linalg.cholesky(A)
    d = np.abs(LP - L1)
    bad = np.argwhere(d > 0)
    print("N=%d P=%d nb=%d rc=%d,%d |L1-ref|=%.2e |LP-ref|=%.2e nbad=%d first bad %s blocks(row,col) %s" % (
        N, P, nb, rc1, rcp, np.abs(L1 - Lr).max(), np.abs(LP - Lr).max(), len(bad), bad[:1].tolist(),
        sorted(set((int(i) // nb, int(j) // nb) for i, j in bad[::max(1, len(bad) // 2000)]))[:12]))
