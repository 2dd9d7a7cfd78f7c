"""Posterior rate at N = 16384 (and 32768) against the two-level blocking parameters."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
from bench import synthetic_problem

_lib.init()
l = _lib.lib()
P = _lib.c_double_p


def run(N, d, M, steps=6, warmup=2):
    X, y, Z, lh = synthetic_problem(N, d, M)
    ms, nl = ctypes.c_double(), ctypes.c_double()
    mean, var = np.zeros(M), np.zeros(M)
    _lib.check(l.gpk_set_option(b"timing", 1.0), "timing")
    rc = l.gpk_bench_posterior(X.ctypes.data_as(P), y.ctypes.data_as(P), N, d, lh.ctypes.data_as(P), Z.ctypes.data_as(P), M,
                               warmup, steps, ctypes.byref(ms), ctypes.byref(nl), mean.ctypes.data_as(P), var.ctypes.data_as(P))
    _lib.check(rc, "bench")
    a, c, w = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    l.gpk_last_timing(b"trailing", ctypes.byref(a), ctypes.byref(c))
    l.gpk_last_timing_work(b"trailing", ctypes.byref(w))
    tf = w.value / (a.value * 1e-3) / 1e12 if a.value else 0
    return ms.value / steps, nl.value, tf, a.value / steps


N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D = 8 if N > 4096 else 4
configs = [dict(nb_outer=0)]
if len(sys.argv) > 2 and sys.argv[2] == "tail":
    for t in (2048, 4096, 6144, 8192, 12288):
        configs.append(dict(tail_rows=t, nb_small_from=t))
    configs.append(dict(tail_rows=1 << 30, nb_small_from=1 << 30))
elif len(sys.argv) > 2 and sys.argv[2] == "tail_main":  # the tile pipeline's full-height GEMMs on the update stream (three streams)
    for t in (1024, 2048, 3072, 4096, 6144):
        configs.append(dict(tail_stream=0, tail_rows=t, nb_small_from=t))
    configs.append(dict(tail_stream=0, tail_rows=2048))
    configs.append(dict(tail_stream=1, tail_rows=2048, nb_small_from=2048))
elif len(sys.argv) > 2 and sys.argv[2] == "panels":
    for big, until in [(512, 4096), (512, 6144), (512, 12288), (1024, 8192), (1024, 4096), (768, 6144)]:
        for small_from in (2048, 4096):
            configs.append(dict(nb_big=big, nb_big_until=until, nb_small_from=small_from))
elif len(sys.argv) > 2 and sys.argv[2] == "misc":
    configs += [dict(xcd_sq_min_tiles=0), dict(xcd_sq_min_tiles=2048), dict(xcd_sq_min_tiles=1 << 30), dict(pair_update=2), dict(pair_update=0),
                dict(small_tile_below=128), dict(small_tile_below=512), dict(pred_small_below=150), dict(pred_small_below=600),
                dict(nb_big_until=4096), dict(nb_big_until=12288), dict(nb_small_from=4096), dict(nb_small_from=1024),
                dict(nb_outer=2048, outer_until=8192, outer_from=8192), dict(nb_outer=1024, outer_until=4096, outer_from=8192)]
elif len(sys.argv) > 2 and sys.argv[2] == "short":
    configs += [dict(nb_outer=1024, outer_until=4096, outer_from=8192), dict(nb_outer=2048, outer_until=4096, outer_from=8192),
                dict(nb_outer=2048, outer_until=8192, outer_from=8192),
                dict(nb_outer=2048, outer_until=4096, outer_from=8192, nb_big_until=1 << 30)]
else:
    for W in (1024, 2048, 4096):
        for until in (2048, 4096, 8192):
            configs.append(dict(nb_outer=W, outer_until=until, outer_from=8192))
    configs += [dict(nb_outer=2048, outer_until=4096, outer_from=8192, nb_big_until=1 << 30),   # inner 256 everywhere
                dict(nb_outer=2048, outer_until=4096, outer_from=8192, nb_big=1024),
                dict(nb_outer=1536, outer_until=4096, outer_from=8192),
                dict(nb_outer=3072, outer_until=4096, outer_from=8192)]
print("GPK_RESERVE_PER_XCD=%s GPU_MAX_HW_QUEUES=%s" % (os.environ.get("GPK_RESERVE_PER_XCD"), os.environ.get("GPU_MAX_HW_QUEUES")), flush=True)
defaults = {k: _lib.get_option(k) for k in ("nb_outer", "outer_until", "outer_from", "nb_big_until", "nb_big", "tail_rows", "tail_stream", "nb_small_from",
                                             "xcd_sq_min_tiles", "pair_update", "small_tile_below", "pred_small_below")}
for cfg in configs:
    for k, v in defaults.items():
        _lib.set_option(k, v)
    for k, v in cfg.items():
        _lib.set_option(k, v)
    t, nl, tf, trms = run(N, D, 1024, steps=(6 if N > 4096 else 40))
    print("%-70s %7.2f ms  %6.2f /s  nlml %.10f  trailing %.1f TFLOP/s (%.1f ms/step)" % (cfg, t, 1e3 / t, nl, tf, trms), flush=True)
_lib.shutdown(watchdog_s=10.0)
