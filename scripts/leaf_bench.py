import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
us = ctypes.c_double()
for algo in (0, 1):
    _lib.set_option("leaf_algo", algo)
    modes = [(0, "full"), (1, "no elimination"), (2, "no inverse"), (3, "load/store only")]
    if algo >= 1:
        modes += [(2 + 64, "no inverse, no diag factor"), (2 + 16, "no inverse, no row substitution"), (2 + 32, "no inverse, no MFMA update"),
                  (2 + 16 + 32 + 64, "no inverse, barriers only")]
    for mode, name in modes:
        l.gpk_bench_leaf(mode, 50, ctypes.byref(us))
        print("leaf algo %d %-32s %.1f us" % (algo, name, us.value), flush=True)
_lib.shutdown(watchdog_s=10.0)
