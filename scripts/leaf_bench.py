import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
us = ctypes.c_double()
for mode, name in ((0, "full"), (1, "no elimination"), (2, "no inverse"), (3, "load/store only")):
    l.gpk_bench_leaf(mode, 50, ctypes.byref(us))
    print("leaf %-16s %.1f us" % (name, us.value))
