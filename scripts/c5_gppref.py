"""BASELINE config 5: GPpref preference-learning Laplace loop, 8192 pairs (nx = 16384), d = 4."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import GPpref, _lib
from oracle import gppref_oracle as po   # input generator (demo_problem) and the fixed-point check only

_lib.init()
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = 4
x, uvi, y = po.demo_problem(n_train=npairs, seed=1, d=d)
lh = np.log([0.3] * d + [1.0, 0.1])
gp = GPpref.PreferenceGaussianProcess(x, uvi, y, delta_f=1e-6)
for rep in range(2):
    t0 = time.perf_counter()
    f, lml = gp.calc_laplace(lh)
    t1 = time.perf_counter()
    print("nx=%d pairs=%d d=%d: calc_laplace %.1f ms, %d Newton iterations, eps=%g, lml=%.10g" % (
        len(x), npairs, d, 1e3 * (t1 - t0), gp.last_iterations, gp.last_eps, lml))
print("f_error trace:", ["%.3g" % e for e in gp.last_trace])
# fixed point of the Newton map at the mode: f = K * grad log p(y|f)   (size-independent property)
lik = po.PrefProbit()
z = lik.z_k(uvi, f, y)
from scipy.special import ndtr
q = (y * lik._isqrt2sig * po.std_norm_pdf(z) / ndtr(z)).reshape(-1)
g = np.zeros(len(x)); np.add.at(g, uvi[:, 1], q); np.add.at(g, uvi[:, 0], -q)
from gptest_amd.GPr import fit_handle
h, _ = fit_handle(x, None, np.concatenate([lh[:d], [lh[d]], [-np.inf]]), jitter=gp.last_eps, keep_k=True)
Kg = np.empty((len(x), 1))
gc = np.ascontiguousarray(g.reshape(-1, 1))
assert _lib.lib().gpk_kmatvec(h.ptr, gc.ctypes.data_as(_lib.c_double_p), 1, Kg.ctypes.data_as(_lib.c_double_p)) == 0
print("mode fixed point  max|f - K grad| = %.3e  (max|f| = %.3f)" % (np.abs(f - Kg).max(), np.abs(f).max()))
