import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
P = _lib.c_double_p
rng = np.random.RandomState(0)
for (m, n, k, sub, lower) in [(128, 128, 16, 0, 0), (256, 384, 48, 1, 0), (1024, 1024, 256, 1, 1), (384, 128, 128, 0, 0)]:
    A = rng.randn(m, k); B = A.copy() if lower else rng.randn(n, k); C0 = rng.randn(m, n); C = C0.copy()
    rc = l.gpk_test_gemm_nt(A.ctypes.data_as(P), B.ctypes.data_as(P), C.ctypes.data_as(P), m, n, k, sub, lower, 128)
    ref = C0 - A @ B.T if sub else A @ B.T
    if lower:
        mask = np.zeros((m, n), bool)
        for ti in range(m // 128): mask[ti * 128:(ti + 1) * 128, :(ti + 1) * 128] = True
        err = np.abs(C - ref)[mask].max()
    else:
        err = np.abs(C - ref).max()
    print("g8 m=%d n=%d k=%d sub=%d lower=%d rc=%d err=%.3e" % (m, n, k, sub, lower, rc, err))
ms = ctypes.c_double()
t = (ctypes.c_double * 2)()
l.gpk_bench_mfma_peak2(8, 4, 40000, t); print("calibration: pure MFMA 4 waves/SIMD %.2f TFLOP/s at %.3f GHz" % (t[0], t[1]))
for (m, n, k, lower) in [(16128, 16128, 256, 1), (15872, 15872, 512, 1), (15360, 15360, 1024, 1), (8192, 8192, 4096, 0), (4096, 4096, 256, 1), (1024, 16128, 256, 0), (1024, 15872, 512, 0)]:
    for tile, nm in ((4, "4-wave"), (8, "8-wave"), (1, "pp")):
        if tile == 1 and k not in (256, 512): continue
        rc = l.gpk_bench_gemm_nt(m, n, k, lower, tile, 3, ctypes.byref(ms))
        fl = (m * n * k) if lower else 2.0 * m * n * k
        print("gemm m=%5d n=%5d k=%4d lower=%d %-6s: %.3f ms %.2f TFLOP/s" % (m, n, k, lower, nm, ms.value, fl / ms.value / 1e9))
