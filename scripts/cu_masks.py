import ctypes, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
nm, nl = 4096, 64
a = (ctypes.c_uint * nm)(); b = (ctypes.c_uint * nl)()
rc = l.gpk_debug_cu_masks(nm, a, nl, b)
A = collections.Counter(a); B = collections.Counter(b)
print("cu_reserved", int(_lib.get_option("cu_reserved")), "rc", rc)
print("GEMM stream: %d workgroups on %d distinct (xcc,se,sh,cu)" % (nm, len(A)))
print("leaf stream: %d workgroups on %d distinct:" % (nl, len(B)), ["%05x" % k for k in B])
print("overlap:", ["%05x" % k for k in B if k in A])
byx = collections.Counter(k >> 16 for k in A)
print("GEMM CUs per XCC:", dict(sorted(byx.items())))
