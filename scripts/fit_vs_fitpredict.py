import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gptest_amd import GPr, _lib
from oracle import gpr_oracle as orc
_lib.init()
for o in sys.argv[1:]:
    k, v = o.split("="); _lib.set_option(k, float(v))
for N, d in [(512, 1), (1024, 4), (4096, 4)]:
    x, y, z, lh = orc.synthetic_problem(N, d, M=100)
    for mode in ("fit keep-one", "fit free-first", "fit_predict"):
        ts = []
        h = None
        for i in range(12):
            hyp = lh + 1e-7 * (i + 1)
            t0 = time.perf_counter()
            if mode == "fit keep-one":
                h2, _ = GPr.fit_handle(x, y, hyp); h = h2
            elif mode == "fit free-first":
                h = None
                h, _ = GPr.fit_handle(x, y, hyp)
            else:
                g = GPr.GaussianProcess(hyp, 0, 0, "SE", "zero", "zero", x, y); g.compute_prediction(z)
            ts.append((time.perf_counter() - t0) * 1e3)
        print("N=%d %-15s: first %.2f ms, then %s" % (N, mode, ts[0], " ".join("%.2f" % t for t in ts[1:])), flush=True)
_lib.shutdown(watchdog_s=10.0)
