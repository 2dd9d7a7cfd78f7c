import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
a, c = ctypes.c_double(), ctypes.c_double()
print("GPK_RESERVE_CU=%s GPK_RESERVE_PER_XCD=%s GPU_MAX_HW_QUEUES=%s leaf_server=%d" % (os.environ.get("GPK_RESERVE_CU"), os.environ.get("GPK_RESERVE_PER_XCD"), os.environ.get("GPU_MAX_HW_QUEUES"), _lib.get_option("leaf_server")))
for nsmall, m in [(0, 128), (2, 128), (3, 256), (2, 4096), (2, 8192)]:
    _lib.check(l.gpk_bench_chain_step(200, nsmall, m, ctypes.byref(a), ctypes.byref(c)), "chain")
    print("leaf + %d GEMMs of %5d x 128 x 128: %6.1f us/step alone, %6.1f us/step beside trailing updates" % (nsmall, m, a.value, c.value), flush=True)
_lib.shutdown(watchdog_s=10.0)
