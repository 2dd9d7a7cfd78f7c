"""Timing-only variants of the 8-wave GEMM (need a library built with `make -C gptest_amd/csrc DBG=1`): GPK_GEMM_DBG bits
1 no global loads / LDS writes in the loop, 2 no barriers, 4 no LDS reads (plain loop), 8 no C load / store, 16 every tile
reads tile 0's first slab; GPK_GEMM_DYNLDS=20480 one workgroup per CU; GPK_BENCH_UNMASKED=1 the unmasked stream; option
gemm_pipe through GPK_GEMM_PIPE=0|1.  Usage: python scripts/gemm_dbg.py [m [k ...]]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
if os.environ.get("GPK_GEMM_PIPE") is not None:
    _lib.set_option("gemm_pipe", float(os.environ["GPK_GEMM_PIPE"]))
ms = ctypes.c_double()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 7936
ks = [int(a) for a in sys.argv[2:]] or [2048, 512, 256]
for k in ks:
    reps = max(4, 12288 // k)
    rc = l.gpk_bench_gemm_nt(m, 8192, k, 0, 8, reps, ctypes.byref(ms))
    rc = l.gpk_bench_gemm_nt(m, 8192, k, 0, 8, reps, ctypes.byref(ms))
    print("DBG=%s reserve=%s unmasked=%s m=%d n=8192 k=%d: %.3f ms %.2f TFLOP/s" % (os.environ.get("GPK_GEMM_DBG", "0"), os.environ.get("GPK_RESERVE_PER_XCD"), os.environ.get("GPK_BENCH_UNMASKED"), m, k, ms.value, 2.0 * m * 8192 * k / ms.value / 1e9), flush=True)
_lib.shutdown()
