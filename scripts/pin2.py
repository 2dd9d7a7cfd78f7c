import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
o = (ctypes.c_double * 6)()
for w, it in ((4, 200000), (4, 200000)):
    rc = l.gpk_bench_mfma_pinned(w, it, o)
    assert rc == 0, _lib.last_error()
    print("pinned W=%d threads=%s masked=%s reserve=%s cus=%s: %6.2f TFLOP/s (events) %6.2f (slowest wave) at %.3f GHz | per SIMD max %d min %d, %d SIMDs" % (w, os.environ.get("GPK_BENCH_THREADS"), os.environ.get("GPK_BENCH_MASKED"), os.environ.get("GPK_RESERVE_PER_XCD"), os.environ.get("GPK_BENCH_CUS"), o[0], o[5], o[1], o[2], o[3], o[4]), flush=True)
_lib.shutdown()
