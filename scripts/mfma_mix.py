import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
t = ctypes.c_double()
for w in (2, 4):
    for iters in (500, 2000, 8000, 40000, 200000):
        l.gpk_bench_mfma_mix(0, w, iters, ctypes.byref(t))
        print("waves/SIMD=%d iters=%6d (%.2f ms at 64 cyc/MFMA, 2.4GHz): %6.2f TFLOP/s" % (w, iters, iters * 8 * 64 * w / 2.4e6, t.value))
ms = ctypes.c_double()
for (m, n, k, lower, tile, reps) in [(8192, 8192, 4096, 0, 128, 1), (8192, 8192, 4096, 0, 128, 10), (8192, 8192, 256, 0, 128, 1), (8192, 8192, 256, 0, 128, 50)]:
    rc = l.gpk_bench_gemm_nt(m, n, k, lower, tile, reps, ctypes.byref(ms))
    print("old gemm m=%d n=%d k=%d reps=%d: %.3f ms %.2f TFLOP/s" % (m, n, k, reps, ms.value, 2.0 * m * n * k / ms.value / 1e9))
