"""GEMM / MFMA-peak micro-benchmarks (device-resident)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpk_hooks import hooks  # libgpk_hooks.so: probes and test hooks live outside the product library
_lib.init()
l = _lib.lib()
t = (ctypes.c_double * 6)()
for it in (20000, 100000):
    l.gpk_bench_mfma_pinned(4, it, t)
    print("mfma f64 16x16x4 issue loop, 4 waves pinned per SIMD: %.2f TFLOP/s at %.3f GHz (iters=%d)" % (t[0], t[1], it))
ms = ctypes.c_double()
for (m, n, k, lower, tile) in [(8192, 8192, 128, 0, 128), (8192, 8192, 256, 0, 128), (8192, 8192, 512, 0, 128), (8192, 8192, 1024, 0, 128),
                               (8192, 8192, 4096, 0, 128), (16128, 16128, 256, 1, 128), (16128, 16128, 512, 1, 128), (8192, 8192, 256, 1, 128),
                               (16128, 128, 128, 0, 64), (16128, 128, 128, 0, 128), (16128, 256, 256, 0, 128), (1024, 16128, 256, 0, 128)]:
    rc = hooks().gpk_bench_gemm_nt(m, n, k, lower, tile, 5, ctypes.byref(ms))
    fl = (m * n * k) if lower else 2.0 * m * n * k
    print("gemm m=%d n=%d k=%d lower=%d tile=%d rc=%d: %.3f ms  %.2f TFLOP/s" % (m, n, k, lower, tile, rc, ms.value, fl / ms.value / 1e9))
