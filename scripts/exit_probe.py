import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
us = ctypes.c_double()
algos = [int(a) for a in sys.argv[1].split(",")]
modes = [int(a) for a in sys.argv[2].split(",")]
for algo in algos:
    _lib.set_option("leaf_algo", algo)
    for mode in modes:
        l.gpk_bench_leaf(mode, 50, ctypes.byref(us))
        print("leaf algo %d mode %d %.1f us" % (algo, mode, us.value), flush=True)
t0 = time.time(); rc = l.gpk_sync(); print("sync rc=%d took %.2f s" % (rc, time.time() - t0), flush=True)
t0 = time.time(); _lib.shutdown(); print("shutdown took %.2f s" % (time.time() - t0), flush=True)
