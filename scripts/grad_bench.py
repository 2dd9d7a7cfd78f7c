"""Time gpk_nlml_grad on the BASELINE configs[1] training set (factor already resident)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gptest_amd import _lib, GPr
from oracle import gpr_oracle as orc   # synthetic inputs only (scripts/ is not the product)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
x, y, _, lh = orc.synthetic_problem(N, d, M=4)
_lib.init()
_lib.set_option("timing", 1)
gp = GPr.GaussianProcess(lh, 0, 0, "SE", "zero", "zero", x, y)
gp.compute_likelihood(lh)
for it in range(3):
    t0 = time.perf_counter()
    nlml, g = gp.compute_likelihood_gradient(lh)
    dt = (time.perf_counter() - t0) * 1e3
    ms, cnt, wk = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    line = "grad N=%d d=%d: %.2f ms wall" % (N, d, dt)
    for st in (b"grad_inv", b"grad_terms"):
        _lib.lib().gpk_last_timing(st, ctypes.byref(ms), ctypes.byref(cnt))
        _lib.lib().gpk_last_timing_work(st, ctypes.byref(wk))
        line += "  %s %.2f ms" % (st.decode(), ms.value)
        if wk.value > 0 and ms.value > 0:
            line += " (%.1f TFLOP/s)" % (wk.value / ms.value / 1e9)
    print(line)
print("grad", g)
if N <= 4096:
    ref = orc.nlml_gradient(lh, x, y)
    print("max rel err vs oracle", np.abs(g - ref).max() / np.abs(ref).max())
