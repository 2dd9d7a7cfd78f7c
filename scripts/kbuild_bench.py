import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpk_hooks import hooks  # libgpk_hooks.so: probes and test hooks live outside the product library
_lib.init()
l = _lib.lib()
ms = ctypes.c_double()
N = 16384
for rep in range(3):
  for d in (8, 4):
    for lower in (1, 0):
      for mode, nm in ((0, "SE"), (3, "SE scalar"), (4, "SE scalar, no pre-pass"), (1, "sqdist"), (2, "stores only")):
            hooks().gpk_bench_se_build(N, d, mode, lower, 50, ctypes.byref(ms))
            byts = (4.0 if lower else 8.0) * N * N
            print("kbuild N=%d d=%d %-5s %-22s: %.3f ms  %.2f TB/s written" % (N, d, "lower" if lower else "full", nm, ms.value, byts / ms.value / 1e9))
