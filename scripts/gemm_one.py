import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
m, n, k, lower, tile, reps = (int(x) for x in sys.argv[1:7])
ms = ctypes.c_double()
rc = l.gpk_bench_gemm_nt(m, n, k, lower, tile, reps, ctypes.byref(ms))
fl = (m * n * k) if lower else 2.0 * m * n * k
print("gemm m=%d n=%d k=%d lower=%d tile=%d rc=%d: %.3f ms  %.2f TFLOP/s" % (m, n, k, lower, tile, rc, ms.value, fl / ms.value / 1e9))
