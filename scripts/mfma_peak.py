import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
_lib.init()
l = _lib.lib()
t = (ctypes.c_double * 2)()
for w in (1, 2, 4):
    for nacc in (1, 4, 8, 16):
        l.gpk_bench_mfma_peak2(nacc, w, 40000, t)
        peak = 256 * 4 * 32 * t[1] * 1e9 / 1e12
        print("f64 mfma 16x16x4: waves/SIMD=%d acc/wave=%2d : %6.2f TFLOP/s  shader clock %.3f GHz -> %.1f%% of the %.1f TFLOP/s peak at that clock" % (w, nacc, t[0], t[1], 100 * t[0] / peak, peak))
ms = ctypes.c_double()
for (m, n, k, lower, tile) in [(8192, 8192, 4096, 0, 128), (16128, 16128, 256, 1, 128), (16128, 16128, 256, 1, 1), (16128, 16128, 512, 1, 128)]:
    rc = l.gpk_bench_gemm_nt(m, n, k, lower, tile, 3, ctypes.byref(ms))
    fl = (m * n * k) if lower else 2.0 * m * n * k
    print("gemm m=%d n=%d k=%d lower=%d %s: %.3f ms %.2f TFLOP/s" % (m, n, k, lower, "pp" if tile == 1 else "old", ms.value, fl / ms.value / 1e9))
