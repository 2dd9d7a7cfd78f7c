import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
us = ctypes.c_double()
for algo in (0, 1):
    _lib.set_option("leaf_algo", algo)
    for mode, name in ((0, "full"), (1, "no elimination"), (2, "no inverse"), (3, "load/store only")):
        l.gpk_bench_leaf(mode, 50, ctypes.byref(us))
        print("leaf algo %d %-16s %.1f us" % (algo, name, us.value))
import time
sys.stdout.flush()
t0 = time.time()
_lib.shutdown()
print("shutdown took %.2f s" % (time.time() - t0), flush=True)
