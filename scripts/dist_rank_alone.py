"""One-GPU measurement of the block-cyclic multi-GPU schedule: every rank of a P-rank posterior replayed ALONE
(gpk_bench_dist_rank_alone): what each rank has to do when it never waits for a peer.  max over ranks = a lower bound of the
job's time on P GPUs.  usage: dist_rank_alone.py N d M P[,P...] [option=value ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gptest_amd import _lib
from gpk_hooks import hooks
from bench import synthetic_problem

N, d, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
Ps = [int(x) for x in sys.argv[4].split(",")]
_lib.init()
comm = 0
for o in sys.argv[5:]:
    k, v = o.split("=")
    if k == "comm_stream":   # the copies on a communication stream of their own (a real rank's four-stream layout)
        comm = int(v)
    else:
        _lib.set_option(k, float(v))
X, y, Z, lh = synthetic_problem(N, d, M)
P_ = _lib.c_double_p
ptr = lambda a: a.ctypes.data_as(P_)
ms1 = ctypes.c_double(); nl = ctypes.c_double(); mean = np.zeros(M); var = np.zeros(M)
_lib.lib().gpk_bench_posterior(ptr(X), ptr(y), N, d, ptr(lh), ptr(Z), M, 1, 4, ctypes.byref(ms1), ctypes.byref(nl), ptr(mean), ptr(var))
print("N=%d d=%d M=%d %s: one GPU %.3f ms/posterior" % (N, d, M, " ".join(sys.argv[5:]), ms1.value / 4))
for P in Ps:
    out = np.zeros(P)
    rc = hooks().gpk_bench_dist_rank_alone(ptr(X), ptr(y), N, d, ptr(lh), ptr(Z), M, P, 3, comm, ptr(out))
    assert rc == 0, _lib.last_error()
    print("  P=%d tail block from panel %d (column %d): each rank alone (ms): %s  -> max %.3f ms = %.2fx of one GPU" % (
        P, _lib.get_option("dist_tail_from"), _lib.get_option("dist_tail_col"), " ".join("%.2f" % v for v in out), out.max(), ms1.value / 4 / out.max()))
_lib.shutdown()
