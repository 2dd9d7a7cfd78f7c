import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpk():
    """The loaded + initialised product library.  GPU tests FAIL (not skip) when it is
    missing: a silent fallback would void every parity claim."""
    from gptest_amd import _lib

    _lib.init()
    return _lib


def golden_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("gpr_") and f.endswith(".npz"))


# every option gpk_set_option knows, in an order that restores correctly ("nb" switches nb_auto off, so nb_auto
# comes after it)
OPTION_KEYS = ["nb", "nb_auto", "nb_big", "nb_big_until", "nb_small", "nb_small_from", "nb_outer", "outer_from",
               "outer_until", "lookahead", "dist_group", "ldk_pad", "pred_gate_rows", "gemm_variant",
               "xcd_sq", "xcd_sorted", "xcd_sq_min_tiles", "tail_rows", "tail_stream", "tt_fused", "tail_leaf_event", "tail_flags", "stamps", "pred_small_below", "small_tile_below", "leaf_algo",
               "test_gemm_k1", "pair_update", "leaf_server", "gemm_pipe", "pipe_interleave", "fuse_predict", "solve_gemm_above", "timing"]


@pytest.fixture(autouse=True)
def _restore_gpk_options(request):
    """GPU tests: snapshot every libgpk option before the test and restore all of them afterwards, so that no test
    can leave another one running a non-default schedule (round-1 advice: 'nb' used to leak nb_auto = 0)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from gptest_amd import _lib

    _lib.init()
    before = {k: _lib.get_option(k) for k in OPTION_KEYS}
    assert all(v == v for v in before.values()), before  # no NaN: every key is readable
    yield
    for k in OPTION_KEYS:
        if _lib.get_option(k) != before[k]:
            _lib.set_option(k, before[k])
    assert {k: _lib.get_option(k) for k in OPTION_KEYS} == before
