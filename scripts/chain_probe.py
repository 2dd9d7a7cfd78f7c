import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpk_hooks import hooks  # libgpk_hooks.so: probes and test hooks live outside the product library
_lib.init()
l = _lib.lib()
a, c = ctypes.c_double(), ctypes.c_double()
for o in sys.argv[1:]:
    k, v = o.split("=")
    _lib.set_option(k, float(v))
print("GPU_MAX_HW_QUEUES=%s leaf_fat_rows=%d gemm_polite=%d" % (os.environ.get("GPU_MAX_HW_QUEUES"), _lib.get_option("leaf_fat_rows"), _lib.get_option("gemm_polite")))
for nsmall, m in [(0, 128), (2, 128), (3, 256), (2, 4096), (2, 8192)]:
    _lib.check(hooks().gpk_bench_chain_step(200, nsmall, m, ctypes.byref(a), ctypes.byref(c)), "chain")
    print("leaf + %d GEMMs of %5d x 128 x 128: %6.1f us/step alone, %6.1f us/step beside trailing updates" % (nsmall, m, a.value, c.value), flush=True)
_lib.shutdown()
