import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
P = _lib.c_double_p
rng = np.random.RandomState(0)
for (m, n, k, lower) in [(128, 128, 16, 0), (256, 384, 48, 0), (1024, 1024, 256, 1), (2176, 2176, 128, 1), (1152, 2688, 64, 0), (5120, 5120, 32, 1)]:
    A = rng.randn(m, k); B = A.copy() if lower else rng.randn(n, k); C0 = rng.randn(m, n); C = C0.copy()
    rc = l.gpk_test_gemm_nt(A.ctypes.data_as(P), B.ctypes.data_as(P), C.ctypes.data_as(P), m, n, k, 1, lower, 128)
    ref = C0 - A @ B.T
    if lower:
        mask = np.zeros((m, n), bool)
        for ti in range(m // 128): mask[ti * 128:(ti + 1) * 128, :(ti + 1) * 128] = True
        err = np.abs(C - ref)[mask].max(); unt = np.abs(C - C0)[~mask].max() if (~mask).any() else 0
    else:
        err = np.abs(C - ref).max(); unt = 0
    print("g8p m=%d n=%d k=%d lower=%d rc=%d err=%.3e untouched=%.1e %s" % (m, n, k, lower, rc, err, unt, _lib.last_error() if rc else ""))
ms = ctypes.c_double()
t = (ctypes.c_double * 2)()
l.gpk_bench_mfma_peak2(8, 4, 40000, t); print("calibration: pure MFMA 4 waves/SIMD %.2f TFLOP/s at %.3f GHz" % (t[0], t[1]))
for (m, n, k, lower) in [(16128, 16128, 256, 1), (15872, 15872, 512, 1), (8192, 8192, 256, 1), (4096, 4096, 256, 1), (2048, 2048, 256, 1), (1024, 16128, 256, 0), (1024, 15872, 512, 0), (1024, 8192, 256, 0)]:
    for tile, nm in ((4, "4-wave (64x64 wave tiles)"), (8, "8-wave"), (10, "8-wave xcd-squares")):
        rc = l.gpk_bench_gemm_nt(m, n, k, lower, tile, 3, ctypes.byref(ms))
        fl = (m * n * k) if lower else 2.0 * m * n * k
        print("gemm m=%5d n=%5d k=%4d lower=%d %-18s: %.3f ms %.2f TFLOP/s" % (m, n, k, lower, nm, ms.value, fl / ms.value / 1e9))
