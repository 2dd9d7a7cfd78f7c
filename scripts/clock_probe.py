"""Is the bulk update clock-throttled?  Shader clock sampled by a one-wave kernel while the update stream runs
back-to-back trailing updates for ~0.3-0.5 s (no rocm-smi: it hangs on the pool's boxes)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
t = (ctypes.c_double * 6)()
l.gpk_bench_mfma_pinned(4, 400000, t)
print("issue loop, 4 waves pinned per SIMD: %.2f TFLOP/s at %.3f GHz (in-kernel clock)" % (t[0], t[1]), flush=True)
ns = 24
ghz = (ctypes.c_double * (ns + 1))()
tf = ctypes.c_double()
for (m, k, nbig) in [(16128, 256, 300), (16128, 512, 160), (16128, 2048, 40), (8192, 256, 600)]:
    _lib.check(l.gpk_bench_clock_under_load(m, k, nbig, ns, ghz, ctypes.byref(tf)), "clock probe")
    g = list(ghz)
    print("lower update m=%d K=%d x%d: %.2f TFLOP/s sustained | clock idle %.3f GHz, under load min %.3f / mean %.3f / max %.3f GHz" % (
        m, k, nbig, tf.value, g[0], min(g[1:]), sum(g[1:]) / ns, max(g[1:])), flush=True)
_lib.shutdown()
