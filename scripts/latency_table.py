"""Per-call latency of the drop-in surface vs the NumPy restatement on the host, from the reference's demo sizes up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gptest_amd import GPr, _lib
from oracle import gpr_oracle as orc
_lib.init()
print("%6s %3s | %12s %12s | %12s %12s" % ("N", "d", "gpu nlml ms", "gpu pred ms", "numpy nlml ms", "numpy pred ms"))
for N, d in [(20, 1), (128, 2), (512, 1), (1024, 4), (2048, 4), (4096, 4), (8192, 8)]:
    x, y, z, lh = orc.synthetic_problem(N, d, M=100)
    xin = x[:, 0] if d == 1 else x
    zin = z[:, 0] if d == 1 else z
    gp = GPr.GaussianProcess(lh, 0, 0, "SE", "zero", "zero", xin, y)
    reps = 20 if N <= 2048 else 5
    for w in range(3):
        gp.compute_likelihood(lh + 1e-9 * (w + 1))   # two handles are alive at a time: warm both sets of buffers
    GPr.GaussianProcess(lh + 1e-9, 0, 0, "SE", "zero", "zero", xin, y).compute_prediction(zin)
    t0 = time.perf_counter()
    for i in range(reps):
        gp.compute_likelihood(lh + 1e-7 * (i + 1))     # new hyper-parameters every call: no cache hit
    t_l = (time.perf_counter() - t0) / reps * 1e3  # mean; first-use allocations of a new size show up here
    t0 = time.perf_counter()
    for i in range(reps):
        g2 = GPr.GaussianProcess(lh + 1e-7 * (i + 1), 0, 0, "SE", "zero", "zero", xin, y)
        g2.compute_prediction(zin)
    t_p = (time.perf_counter() - t0) / reps * 1e3
    if N <= 4096:
        r2 = 3 if N <= 1024 else 1
        t0 = time.perf_counter()
        for i in range(r2): orc.compute_likelihood_verbatim(lh, x, y)
        c_l = (time.perf_counter() - t0) / r2 * 1e3
        t0 = time.perf_counter()
        for i in range(r2): orc.compute_prediction_verbatim(lh, x, y, z)
        c_p = (time.perf_counter() - t0) / r2 * 1e3
        print("%6d %3d | %12.3f %12.3f | %12.3f %12.3f" % (N, d, t_l, t_p, c_l, c_p), flush=True)
    else:
        print("%6d %3d | %12.3f %12.3f |" % (N, d, t_l, t_p), flush=True)
