"""A/B of the stream layouts on one box: posterior / factorisation-only times at a few sizes.
Run once per mode:  python scripts/mode_ab.py           (resident leaf server, nothing masked)
                    GPK_RESERVE_CU=1 python scripts/mode_ab.py   (CU-masked update stream, direct leaf launches)"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib  # noqa: E402
from bench import synthetic_problem  # noqa: E402

_lib.init()
l = _lib.lib()
for o in sys.argv[1:]:
    k, v = o.split("=")
    _lib.set_option(k, float(v))
P = _lib.c_double_p
print("mode: GPK_RESERVE_CU=%s leaf_server=%d cu_reserved=%d opts=%s" % (os.environ.get("GPK_RESERVE_CU"), _lib.get_option("leaf_server"), _lib.get_option("cu_reserved"), sys.argv[1:]), flush=True)
for (N, d, M, reps) in [(16384, 8, 1024, 10), (16384, 8, 128, 10), (4096, 4, 1024, 40), (8192, 8, 1024, 20), (32768, 8, 1024, 3)]:
    X, y, Z, lh = synthetic_problem(N, d, max(M, 1))
    ms = ctypes.c_double()
    nl = ctypes.c_double()
    mean = np.zeros(max(M, 1))
    var = np.zeros(max(M, 1))
    for rep in range(2):
        rc = l.gpk_bench_posterior(X.ctypes.data_as(P), y.ctypes.data_as(P), N, d, lh.ctypes.data_as(P), Z.ctypes.data_as(P), M, 1, reps,
                                   ctypes.byref(ms), ctypes.byref(nl), mean.ctypes.data_as(P), var.ctypes.data_as(P))
        assert rc == 0, _lib.last_error()
    print("N=%5d d=%d M=%4d: %8.3f ms  nlml=%.12g" % (N, d, M, ms.value / reps, nl.value), flush=True)
_lib.shutdown(watchdog_s=10.0)
