"""Does the guest poll cost the BULK kernel anything?  gpk_bench_gemm_nt (back-to-back launches, alone) with gemm_polite = 1 / 0, alternating."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpk_hooks import hooks
_lib.init()
ms = ctypes.c_double()
for (m, n, k, lower, tile) in [(16128, 16128, 512, 1, 128), (8192, 8192, 256, 1, 128), (16128, 16128, 2048, 1, 128), (16128, 128, 128, 0, 64), (8064, 256, 256, 0, 64), (1024, 4096, 128, 0, 64)]:
    for rnd in range(3):
        for pol in (1, 0):
            _lib.set_option("gemm_polite", pol)
            hooks().gpk_bench_gemm_nt(m, n, k, lower, tile, 20, ctypes.byref(ms))
            fl = (m * n * k) if lower else 2.0 * m * n * k
            print("m=%d n=%d k=%d lower=%d tile=%d polite=%d: %.4f ms  %.2f TFLOP/s" % (m, n, k, lower, tile, pol, ms.value, fl / ms.value / 1e9), flush=True)
_lib.shutdown()
