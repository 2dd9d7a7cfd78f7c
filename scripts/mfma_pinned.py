"""fp64 MFMA issue loop with the residency pinned (exactly W waves on every SIMD, through the LDS allocation).
Environment: GPK_BENCH_CUS=<n> sizes the grid for n CUs, GPK_BENCH_THREADS=512|1024 changes the workgroup size (W = 4 only).
(The dispatcher-placed loop and the CU-masked stream it was compared with in round 2 are gone: profiles/r02_mfma_pinned_and_mask.txt.)
Usage: python scripts/mfma_pinned.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib

_lib.init()
l = _lib.lib()
o = (ctypes.c_double * 6)()
for w in (1, 2, 4, 8):
    for rep in range(2):
        rc = l.gpk_bench_mfma_pinned(w, 40000, o)
        assert rc == 0, _lib.last_error()
        peak = 256 * 4 * 32 * o[1] * 1e-3  # TFLOP/s at the measured shader clock
        print("pinned   W=%d: %6.2f TFLOP/s (events) %6.2f (slowest wave's cycles) at %.3f GHz = %.1f%% / %.1f%% of %.1f | waves per SIMD max %d min %d, %d SIMDs"
              % (w, o[0], o[5], o[1], 100 * o[0] / peak, 100 * o[5] / peak, peak, o[2], o[3], o[4]), flush=True)
_lib.shutdown()
