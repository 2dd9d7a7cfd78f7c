"""Host-side sanitizer build: the HOST orchestration of libgpk -- panel schedules, block-cyclic
phases on virtual ranks, pool allocator, Laplace / append drivers, error paths -- compiled as plain C++ against a
host-memory HIP stand-in and CPU launch_* stand-ins (tests/hostsim) with -fsanitize=address,undefined, and run through
the C-ABI with numerical checks against a small dense reference.  No GPU involved; the product never loads any of it."""
import os
import subprocess

from conftest import ROOT

HOSTSIM = os.path.join(ROOT, "tests", "hostsim")


def test_host_orchestration_under_asan_ubsan():
    subprocess.run(["make", "-C", HOSTSIM, "-j8"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([os.path.join(HOSTSIM, "build", "hostsim_tests")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-4000:]
    assert "all checks passed" in p.stdout
    assert "AddressSanitizer" not in p.stdout and "runtime error" not in p.stdout


def test_product_never_reaches_the_host_simulation():
    """The stand-ins are test infrastructure: nothing under gptest_amd/, bench.py or __graft_entry__.py may name them."""
    for base, _, files in os.walk(os.path.join(ROOT, "gptest_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(base, f), errors="replace").read()
                assert "tests/hostsim" not in txt and "sim_kernels" not in txt, os.path.join(base, f)
    for f in ("bench.py", "__graft_entry__.py"):
        assert "hostsim" not in open(os.path.join(ROOT, f)).read()
