import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
a, c = ctypes.c_double(), ctypes.c_double()
for mode, name in ((8, "full+stamps"),):
    rc = l.gpk_bench_leaf_contended(40 | (mode << 16), ctypes.byref(a), ctypes.byref(c))
    sys.stdout.flush()
    print("leaf %-16s alone %.1f us, with the GEMM stream busy %.1f us" % (name, a.value, c.value))
