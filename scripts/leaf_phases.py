import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpk_hooks import hooks  # libgpk_hooks.so: probes and test hooks live outside the product library
_lib.init()
l = _lib.lib()
us = ctypes.c_double()
for form, fname in [(0, "slot-sized leaf (512 threads)"), (256, "full-tile leaf (1024 threads)")]:
    for skip, what in [(0, "full leaf"), (1, "no elimination"), (2, "no inverse"), (3, "neither (load + stores only)")] + ([(16, "no panel solve inside"), (32, "no trailing tile updates"), (64, "no diagonal blocks")] if form else []):
        for rep in range(2):
            hooks().gpk_bench_leaf(skip | form, 400, ctypes.byref(us))
        print("%s dbg_skip=%2d %-32s %6.2f us per launch (back to back on one stream)" % (fname, skip, what, us.value), flush=True)
_lib.shutdown()
