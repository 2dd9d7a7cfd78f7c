import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
ms = ctypes.c_double()
m = 16128
for dbg, name in ((0, "full"), (1, "no C traffic"), (2, "no A/B loads"), (3, "no C, no A/B loads"), (8, "no LDS writes"), (11, "no C/AB/LDSwrite"), (15, "no C/AB/LDSwrite/barrier")):
    for k in (256,):
        rc = l.gpk_bench_gemm_nt(m, m, k, 1 | (dbg << 1), 1, 3, ctypes.byref(ms))
        print("pp %-28s k=%d: %.3f ms  %.2f TFLOP/s" % (name, k, ms.value, m * m * k / ms.value / 1e9))
