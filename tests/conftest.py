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
