"""Runs the six address patterns of gpk_bench_lds_bank_probe once each (for rocprofv3 --pmc: scripts/lds_bank_probe.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpk_hooks import hooks  # libgpk_hooks.so: probes and test hooks live outside the product library
_lib.init()
for pat in range(6):
    _lib.check(hooks().gpk_bench_lds_bank_probe(pat, 4096), "probe")
_lib.shutdown()
