"""CPU: the C-ABI shared library loads and exports every symbol include/gpk.h declares
and include/gpk_bench.h (the test hooks of gpk_testhooks.h: libgpk_hooks.so; no compute calls: there is no GPU here), and the product path fails loudly without one."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols(headers):
    names = set()
    for h in headers:
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(gpk_[A-Za-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from gptest_amd import _lib

    l = _lib.lib()
    names = _declared_symbols(("gpk.h", "gpk_bench.h"))  # the API, and bench.py's measurement entry points exported next to it
    assert len(names) >= 30
    for n in names:
        assert hasattr(l, n), "libgpk.so does not export %s" % n
    # and the Python binding table covers the headers (no silently unbound entry points)
    assert set(names) == set(_lib.EXPORTS.keys())
    assert l.gpk_abi_version() == 1


def test_hooks_live_in_their_own_library():
    """include/gpk_testhooks.h is exported by libgpk_hooks.so (tests/ and scripts/ only), not by the product library."""
    import subprocess

    from gptest_amd import _lib
    import gpk_hooks

    h = gpk_hooks.hooks()
    names = set(_declared_symbols(("gpk_testhooks.h",))) - set(_declared_symbols(("gpk.h", "gpk_bench.h")))
    assert len(names) >= 10
    for n in names:
        assert hasattr(h, n), "libgpk_hooks.so does not export %s" % n
    assert names == set(gpk_hooks.EXPORTS.keys())
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib._LIB_PATH], check=True, stdout=subprocess.PIPE, text=True).stdout
    exported = set(re.findall(r"\b(gpk_[A-Za-z0-9_]+)$", syms, flags=re.M))
    assert not [n for n in exported if n.startswith("gpk_test_")], "the product library exports test hooks"
    assert not (exported & names), exported & names
    # nothing in the product, bench.py or the smoke entry loads the hooks library
    for f in ("bench.py", "__graft_entry__.py"):
        assert "gpk_hooks" not in open(os.path.join(ROOT, f)).read().replace("libgpk_hooks.so", "")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gptest_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "gpk_hooks" not in open(os.path.join(dirpath, f)).read(), f


def test_no_cpu_fallback_without_gpu():
    from gptest_amd import _lib, GPr

    if _lib.lib().gpk_device_count() > 0:
        pytest.skip("a GPU is visible; the no-device failure mode cannot be exercised")
    gp = GPr.GaussianProcess(np.log([1, 1, 0.1]), 0, 0, "SE", "zero", "zero", np.linspace(0, 1, 5), np.zeros(5))
    with pytest.raises(_lib.GpkError):
        gp.compute_likelihood(np.log([1, 1, 0.1]))
    with pytest.raises(_lib.GpkError):
        GPr.squared_distance(np.zeros(3), np.zeros(3))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under gptest_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gptest_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), "%s imports the oracle" % f
                assert "import_module(\"oracle" not in src and "__import__(\"oracle" not in src
                assert "/root/reference" not in src


def test_gpr_surface_matches_reference_names():
    """Constructor attributes and class names of GPr.py:16-42,72-97 (no GPU needed)."""
    from gptest_amd import GPr

    lh = np.log([0.5, 2.0, 0.1])
    x = np.linspace(0, 1, 7)
    y = np.sin(x)
    gp = GPr.GaussianProcess(lh, 0, 0, "SE", "zero", "zero", x, y)
    for a in ("log_hyp", "mean_hyp", "like_hyp", "covFunName", "meanFunName", "likeFunName", "trainInput",
              "trainTarget", "covFun", "meanFun", "likeFun"):
        assert hasattr(gp, a)
    cf = gp.covFun
    assert isinstance(cf, GPr.SquaredExponential) and isinstance(cf, GPr.CovarianceFunction)
    np.testing.assert_allclose(cf.M, [0.5])
    assert cf.sf2 == pytest.approx(4.0) and cf.sn2 == pytest.approx(0.01)
    assert cf.x is x and cf.logHyp is lh
    assert gp.meanFun.y == 0 and gp.likeFun.y == 1.0
    # unknown names silently give [] exactly like GPr.py:31-42
    gp2 = GPr.GaussianProcess(lh, 0, 0, "XX", "YY", "ZZ", x, y)
    assert gp2.covFun == [] and gp2.meanFun == [] and gp2.likeFun == []


def test_traffic_model_matches_the_committed_bench_line():
    """scripts/pmc_traffic_json.py restates the default panel schedule to price the algorithmic bytes of the trailing
    launches; its flop total must be the one bench.py measured from the library's own launch accounting (the committed
    bench line), or the traffic ratio in DESIGN.md compares two different sets of launches."""
    import importlib.util
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pmc_traffic_json", os.path.join(root, "scripts", "pmc_traffic_json.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    model = mod.alg_model()
    import glob

    newest = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench_line.json")))[-1]  # the round that set the defaults
    with open(newest) as fh:
        line = json.loads(fh.read())
    r = line["roofline"]
    assert len(model) * line["steps"] == r["launches"]
    assert sum(f for _, f in model) == r["alg_flops_per_step_in_timed_launches"]
    with open(newest.replace("_bench_line.json", "_traffic_gemm_nt8.json")) as fh:
        tj = json.load(fh)
    assert tj["launches"] == len(model) == tj["launches_expected_from_schedule"]
