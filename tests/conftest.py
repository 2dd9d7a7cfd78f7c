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


# every option gpk_set_option knows (schedule switches, the launch-graph switch, the two diagnostics)
OPTION_KEYS = ["nb", "nb_big_until", "nb_small_from", "nb_outer", "outer_until", "lookahead", "dist_group", "dist_pipeline", "dist_tail", "dist_tail_chunk", "dist_tail_alone", "se_scalar", "kbuild_split", "tail_stream", "tail_rows",
               "leaf_fat_rows", "gemm_polite", "pair_update", "fuse_predict", "small_tile_below", "pred_defer_from", "tail_tile_kernel", "chain_fused",
               "corner_from", "corner_from2", "corner_from3", "corner_chunk", "corner_last", "corner_prio", "corner_hold", "corner_grid", "stamps", "timing"]


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
