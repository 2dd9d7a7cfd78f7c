"""Does the null stream count against the three concurrent streams?  n streams of which the LAST is the null stream
(gpk_bench_streams, use_lib_streams bit 1), with and without the library's own three streams in the set (bit 0)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gptest_amd import _lib
from gpk_hooks import hooks
_lib.init()
out = (ctypes.c_double * 9)()
for use in (2, 3, 0, 1):
    for n in (3, 4, 5):
        for hm, lm, tag in ((0, 0, "all normal"), (0xff, 0, "all high"), (2, 0, "stream 1 high")):
            rc = hooks().gpk_bench_streams(n, hm, lm, 64, 10, 200, use, out)
            assert rc == 0, _lib.last_error()
            print("lib_streams=%d last_is_null=%d n=%d %-14s per-launch us: %s | wall/round %.1f" % (use & 1, (use >> 1) & 1, n, tag, " ".join("%5.1f" % out[i] for i in range(n)), out[n]))
_lib.shutdown()
