"""Diagnostic (non-profiled): when does the substitution stream finish panel p's work relative to the panel chain?"""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gptest_amd import _lib
from oracle.gpr_oracle import synthetic_problem
N, d, M = 16384, 8, 1024
args = sys.argv[1:]
if args and args[0].isdigit():
    N = int(args.pop(0)); d = int(args.pop(0))
_lib.init()
l = _lib.lib()
for o in args:
    k, v = o.split("=")
    _lib.set_option(k, float(v))
_lib.set_option("stamps", 1)
X, y, Z, lh = synthetic_problem(N, d, M)
P = _lib.c_double_p
ms = ctypes.c_double(); nl = ctypes.c_double(); mean = np.zeros(M); var = np.zeros(M)
l.gpk_bench_posterior(X.ctypes.data_as(P), y.ctypes.data_as(P), N, d, lh.ctypes.data_as(P), Z.ctypes.data_as(P), M, 1, 4,
                      ctypes.byref(ms), ctypes.byref(nl), mean.ctypes.data_as(P), var.ctypes.data_as(P))
print("%.3f ms/posterior" % (ms.value / 4))
rows = []
p = 0
while True:
    a = _lib.get_option("stamp_panel_%d" % p); b = _lib.get_option("stamp_subst_%d" % p)
    if math.isnan(a): break
    rows.append((p, a, b)); p += 1
print("panel: chain-done ms / substitutions-done ms / lag")
print("  ".join("%d: %.1f/%.1f/%.1f" % (p, a, b, b - a) for p, a, b in rows[::max(1, len(rows) // 20)] + rows[-1:]))
_lib.shutdown(watchdog_s=10.0)
