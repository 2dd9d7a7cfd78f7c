"""Timing-only variants of the 8-wave GEMM (need a library built with `make -C gptest_amd/csrc DBG=1`): GPK_GEMM_DBG bits
1 no DMA in the loop, 2 no barriers, 8 no C load / store, 16 every tile reads tile 0's first slab; GPK_GEMM_DYNLDS=40960 one
workgroup per CU.  Usage: python scripts/gemm_dbg.py [m [k ...]]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpk_hooks import hooks  # libgpk_hooks.so: probes and test hooks live outside the product library
_lib.init()
l = _lib.lib()
ms = ctypes.c_double()
if os.environ.get("GPK_POLITE") is not None:
    l.gpk_set_option(b"gemm_polite", ctypes.c_double(float(os.environ["GPK_POLITE"])))
m = int(sys.argv[1]) if len(sys.argv) > 1 else 7936
ks = [int(a) for a in sys.argv[2:]] or [2048, 512, 256]
for k in ks:
    reps = max(4, 12288 // k)
    rc = hooks().gpk_bench_gemm_nt(m, 8192, k, 0, 128, reps, ctypes.byref(ms))
    rc = hooks().gpk_bench_gemm_nt(m, 8192, k, 0, 128, reps, ctypes.byref(ms))
    print("polite=%s DBG=%s m=%d n=8192 k=%d: %.3f ms %.2f TFLOP/s" % (os.environ.get("GPK_POLITE", "1"), os.environ.get("GPK_GEMM_DBG", "0"), m, k, ms.value, 2.0 * m * 8192 * k / ms.value / 1e9), flush=True)
_lib.shutdown()
