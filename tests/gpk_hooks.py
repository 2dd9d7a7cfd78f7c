"""TEST / MEASUREMENT INFRASTRUCTURE: ctypes binding of libgpk_hooks.so (include/gpk_testhooks.h).

The hooks -- single kernels behind host buffers, the single-process emulation of the block-cyclic job, hardware
probes -- are a library of their own on top of libgpk.so (gptest_amd/csrc/gpk_hooks.hip).  Only tests/ and scripts/
load it; nothing under gptest_amd/, bench.py or __graft_entry__.smoke() does, and the product library exports no
gpk_test_* symbol (tests/test_abi.py).
"""
import ctypes
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(ROOT, "gptest_amd", "libgpk_hooks.so")
_hooks = None
EXPORTS = None


def _bind(l):
    i, d = ctypes.c_int, ctypes.c_double
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int)
    sig = {
        "gpk_test_rccl_selfcheck": (i, [i]),
        "gpk_test_gemm_nt": (i, [dp, dp, dp, i, i, i, i, i, i, i, i]),
        "gpk_test_gemm_segs": (i, [dp, dp, dp, i, i, ip, ip, i, i]),
        "gpk_test_potf2_128": (i, [dp, dp, dp, i]),
        "gpk_test_chain_step": (i, [dp, dp, dp, dp, i]),
        "gpk_test_dist_potrf": (i, [dp, i, i, dp]),
        "gpk_test_dist_fit": (i, [dp, i, i, dp, d, i, dp]),
        "gpk_test_dist_posterior": (i, [dp, dp, i, i, dp, dp, i, i, dp, dp, dp]),
        "gpk_bench_dist_rank_alone": (i, [dp, dp, i, i, dp, dp, i, i, i, i, dp]),
        "gpk_bench_se_build": (i, [i, i, i, i, i, dp]),
        "gpk_bench_gemm_nt": (i, [i, i, i, i, i, i, dp]),
        "gpk_bench_lds_bank_probe": (i, [i, i]),
        "gpk_bench_streams": (i, [i, i, i, i, i, i, i, dp]),
        "gpk_test_streams_abreast": (i, [i, i, i, dp]),
        "gpk_bench_leaf": (i, [i, i, dp]),
        "gpk_bench_leaf_contended": (i, [i, dp, dp]),
        "gpk_bench_chain_step": (i, [i, i, i, dp, dp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(l, name)  # AttributeError here = header/library mismatch
        f.restype = res
        f.argtypes = args
    return sig


def hooks():
    """Load libgpk_hooks.so (once), after the product library it is linked against."""
    global _hooks, EXPORTS
    if _hooks is None:
        from gptest_amd import _lib

        _lib.lib()  # libgpk.so first (RTLD_GLOBAL): the hooks resolve the library's internal entry points against it
        if not os.path.exists(_PATH):
            raise _lib.GpkError("libgpk_hooks.so is not built (%s): run `make -C gptest_amd/csrc`" % _PATH)
        l = ctypes.CDLL(_PATH, mode=ctypes.RTLD_GLOBAL)
        EXPORTS = _bind(l)
        _hooks = l
    return _hooks
