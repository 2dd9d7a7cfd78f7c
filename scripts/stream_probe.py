"""How many streams run side by side on this runtime?  (gpk_bench_streams: small spinning kernels, round-robin over n streams.)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gptest_amd import _lib
from gpk_hooks import hooks
_lib.init()
out = (ctypes.c_double * 9)()
print("GPU_MAX_HW_QUEUES=%s" % os.environ.get("GPU_MAX_HW_QUEUES"))
for use_lib in (0, 1):
    for n in (1, 2, 3, 4, 5, 6):
        for hm, lm, tag in ((0, 0, "all normal"), (6, 0, "streams 1,2 high"), (6, 8, "1,2 high, 3 low"), (0xff, 0, "all high")):
            rc = hooks().gpk_bench_streams(n, hm, lm, 64, 10, 200, use_lib, out)
            assert rc == 0, _lib.last_error()
            print("lib_streams=%d n=%d %-18s per-launch us: %s | wall/round %.1f" % (use_lib, n, tag, " ".join("%5.1f" % out[i] for i in range(n)), out[n]))
_lib.shutdown()
