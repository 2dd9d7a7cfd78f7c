"""Profiling target: a few posterior solves at one (N, d, M); used under rocprofv3."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib  # noqa: E402
from bench import synthetic_problem  # noqa: E402  (the bench's own input generator)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
M = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
opts = sys.argv[5:]
_lib.init()
l = _lib.lib()
for o in opts:
    k, v = o.split("=")
    _lib.set_option(k, float(v))
if not any(o.startswith("timing=") for o in opts):
    l.gpk_set_option(b"timing", 1.0)
X, y, Z, lh = synthetic_problem(N, d, M)
P = _lib.c_double_p
ms = ctypes.c_double()
nl = ctypes.c_double()
mean = np.zeros(M)
var = np.zeros(M)
rc = l.gpk_bench_posterior(X.ctypes.data_as(P), y.ctypes.data_as(P), N, d, lh.ctypes.data_as(P), Z.ctypes.data_as(P), M, 1, reps,
                           ctypes.byref(ms), ctypes.byref(nl), mean.ctypes.data_as(P), var.ctypes.data_as(P))
print("N=%d d=%d M=%d rc=%d  %.3f ms/posterior  (%.2f posteriors/s) nlml=%.12g" % (N, d, M, rc, ms.value / reps, 1e3 * reps / ms.value, nl.value))
import hashlib
print("   digest of mean | var: %s" % hashlib.sha1(mean.tobytes() + var.tobytes()).hexdigest()[:16])
print("   host time to queue the last factorisation: %.2f ms, the pipelined substitutions: %.2f ms" % (_lib.get_option("potrf_enqueue_ms"), _lib.get_option("pipe_enqueue_ms")))
for st in ("kbuild", "potrf", "panel", "trailing", "solve_y", "predict"):
    a = ctypes.c_double()
    c = ctypes.c_double()
    l.gpk_last_timing(st.encode(), ctypes.byref(a), ctypes.byref(c))
    print("   %-9s %9.3f ms/posterior  (%d launches)" % (st, a.value / reps, int(c.value)))
_lib.shutdown()
