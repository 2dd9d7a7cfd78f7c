"""Device-side stamps (no profiler): per-panel completion times of the chain and of the substitution stream."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gptest_amd import _lib
from bench import synthetic_problem
N, d, M = 4096, 4, 1024
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
l.gpk_bench_posterior(X.ctypes.data_as(P), y.ctypes.data_as(P), N, d, lh.ctypes.data_as(P), Z.ctypes.data_as(P), M, 2, 4,
                      ctypes.byref(ms), ctypes.byref(nl), mean.ctypes.data_as(P), var.ctypes.data_as(P))
print("%s: %.3f ms/posterior; host enqueue potrf %.2f ms, substitutions %.2f ms" % (" ".join(args), ms.value / 4, _lib.get_option("potrf_enqueue_ms"), _lib.get_option("pipe_enqueue_ms")))
rows = []
p = 0
while True:
    a = _lib.get_option("stamp_panel_%d" % p); b = _lib.get_option("stamp_subst_%d" % p)
    if math.isnan(a): break
    rows.append((p, a, b)); p += 1
ch = [a for _, a, _ in rows]
dif = [1e3 * (ch[i] - ch[i - 1]) for i in range(1, len(ch))]
print("panels %d; chain done at %.3f ms; per-panel chain step (us): %s" % (len(rows), ch[-1], " ".join("%.0f" % x for x in dif)))
cr = []
while not math.isnan(_lib.get_option("stamp_corner_%d" % len(cr))):
    cr.append(_lib.get_option("stamp_corner_%d" % len(cr)))
if cr:
    print("lazy corner from column %d: catch-up launches done at (ms): %s" % (_lib.get_option("potrf_corner_from"), " ".join("%.2f" % x for x in cr)))
print("posterior start (K build queued) %.3f ms before panel 0 was factored" % -_lib.get_option("stamp_start"))
print("chain at (ms): %s" % " ".join("%.2f" % x for x in ch))
sb = [b for _, _, b in rows]
print("substitution lag behind the chain (us): %s" % " ".join("%.0f" % (1e3 * (b - a)) for _, a, b in rows))
_lib.shutdown()
