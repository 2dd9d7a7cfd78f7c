"""Host-side sanitizer build: the HOST orchestration of libgpk -- panel schedules, block-cyclic
phases on virtual ranks, pool allocator, Laplace / append drivers, error paths -- compiled as plain C++ against a
host-memory HIP stand-in and CPU launch_* stand-ins (tests/hostsim) with -fsanitize=address,undefined, and run through
the C-ABI with numerical checks against a small dense reference.  No GPU involved; the product never loads any of it."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT

HOSTSIM = os.path.join(ROOT, "tests", "hostsim")


def _build():
    subprocess.run(["make", "-C", HOSTSIM, "-j8"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)


def test_host_orchestration_under_asan_ubsan():
    _build()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([os.path.join(HOSTSIM, "build", "hostsim_tests")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-4000:]
    assert "all checks passed" in p.stdout
    assert "AddressSanitizer" not in p.stdout and "runtime error" not in p.stdout


def test_real_rank_branches_with_two_and_four_processes():
    """The REAL-rank code of the multi-GPU path -- comm_bcast / comm_allgather / gpk_comm_allreduce_max, fit_group and
    predict_group with one local model, the multi-rank leg of gpk_bench_posterior, gpk_comm_init / destroy -- executed by 2
    and by 4 real processes over the inter-process RCCL stand-in (tests/hostsim/sim_rccl.cpp: blocking collectives that
    compare operation, count and root across ranks on every call, bounded waits), under ASan + UBSan.  Every rank's
    nlml / mean / variance equals the dense reference, all ranks agree bit for bit, every job agrees with the one-process
    job to 1e-10, a failing pivot on one owner comes back as the same LAPACK-style code on every rank, a rank whose peers
    never arrive gets an error instead of a hang."""
    _build()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([os.path.join(HOSTSIM, "build", "hostsim_ranks")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-4000:]
    assert "hostsim ranks: all checks passed" in p.stdout
    assert "4 processes agree bit for bit" in p.stdout and "2 processes agree bit for bit" in p.stdout
    assert "AddressSanitizer" not in p.stdout and "runtime error" not in p.stdout


def test_bench_py_as_two_ranks_on_the_cpu():
    """`bench.py --gpus 2` exactly as the driver starts it (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the
    environment, one process per rank) with libgpk's host code running on the CPU stand-ins: argument parsing, rank
    discovery, the TCP bootstrap of the unique id, gpk_comm_init(rank, 2), gpk_bench_posterior's multi-rank branch, the
    barriers and the max over ranks, ONE JSON line from rank 0 -- and that line's nlml is the oracle's."""
    _build()
    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    N, d, M = 700, 3, 50
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--train-n", str(N), "--dim", str(d), "--test-m", str(M)]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HOSTSIM, "run_bench_rank.py")] + args, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d: %s" % (r, outs[r][1][-3000:])
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]  # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "strong"
    assert line["unit"] == "posteriors/s" and line["value"] > 0 and line["dtype"] == "f64"
    assert "cpu_baseline" not in line  # rank 0 at N = 1 only
    from oracle import gpr_oracle as orc

    X, y, Z, lh = orc.synthetic_problem(N, d, M)
    ref = orc.posterior_efficient(lh, X, y, Z)
    nl = float(np.ravel(ref[0])[0])
    assert abs(line["nlml"] - nl) <= 1e-9 * abs(nl), (line["nlml"], nl)


def test_product_never_reaches_the_host_simulation():
    """The stand-ins are test infrastructure: nothing under gptest_amd/, bench.py or __graft_entry__.py may name them."""
    for base, _, files in os.walk(os.path.join(ROOT, "gptest_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(base, f), errors="replace").read()
                assert "tests/hostsim" not in txt and "sim_kernels" not in txt, os.path.join(base, f)
    for f in ("bench.py", "__graft_entry__.py"):
        assert "hostsim" not in open(os.path.join(ROOT, f)).read()
