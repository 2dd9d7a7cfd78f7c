"""CPU, world_size 2 over gloo: the host side of the multi-GPU path (rank discovery, block-cyclic
ownership, unique-id ferrying) and the distributed algorithm itself restated in NumPy with a real
torch.distributed broadcast per panel (oracle/dist_chol_oracle.py)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from gptest_amd import dist


def test_block_cyclic_ownership():
    for world in (1, 2, 3, 8):
        for nblocks in (1, 5, 16, 17):
            for tail_from in (None, 1, nblocks - 2):
                if tail_from is not None and not 0 < tail_from < nblocks:
                    continue
                for alone in (False, True):
                    seen = []
                    for r in range(world):
                        mine = dist.local_blocks(r, world, nblocks, tail_from, alone)
                        assert all(dist.block_cyclic_owner(b, world, tail_from, alone) == r for b in mine)
                        seen += mine
                    assert sorted(seen) == list(range(nblocks))
                    if tail_from is None:
                        continue
                    # the tail block has ONE owner: the cyclic owner of block tail_from, or (alone, 3 ranks or more) the last rank,
                    # which then owns nothing in front of the tail
                    want = world - 1 if (alone and world >= 3) else tail_from % world
                    assert {dist.block_cyclic_owner(b, world, tail_from, alone) for b in range(tail_from, nblocks)} == {want}
                    if alone and world >= 3:
                        assert all(dist.block_cyclic_owner(b, world, tail_from, alone) == b % (world - 1) for b in range(tail_from))


def test_env_rank_parsing(monkeypatch):
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert dist.env_rank() == (3, 1, 4)
    monkeypatch.setenv("RANK", "4")
    with pytest.raises(ValueError):
        dist.env_rank()


WORKER = textwrap.dedent(
    """
    import os, sys
    import numpy as np
    sys.path.insert(0, %r)
    import torch.distributed as tdist
    from gptest_amd import dist
    from oracle import dist_chol_oracle as dco

    rank, local, world = dist.env_rank()
    tdist.init_process_group(backend="gloo", rank=rank, world_size=world)
    # (1) the unique-id bootstrap: rank 0's 128 bytes reach every rank unchanged
    payload = bytes(range(128)) if rank == 0 else None
    got = dist.torch_bootstrap(payload)
    assert got == bytes(range(128)), got[:8]
    # (2) block-cyclic right-looking Cholesky with one broadcast per panel, against LAPACK
    rng = np.random.RandomState(5)
    N, nb = 200, 32
    Q = rng.randn(N, N)
    A = Q @ Q.T + N * np.eye(N)
    ref = np.linalg.cholesky(A)
    for group in (1, 2, 3):   # panels per ownership block (libgpk option dist_group)
        L = dco.cholesky_block_cyclic(A, nb, rank, world, dco.torch_bcast, group=group)
        assert np.allclose(L, ref, rtol=1e-11, atol=1e-11), (group, np.abs(L - ref).max())
    # ... and with the last panels as one tail block on one rank, sent a chunk of panels per broadcast (gpk_model::ct, dist_tail_phase)
    for group, tail_from, chunk in ((1, 3, 1), (1, 4, 2), (2, 4, 8), (3, 3, 3)):
        for alone in ((False, True) if world >= 3 else (False,)):   # (alone: the last rank owns the tail block and nothing else, gpk_model::Pc)
            Lt = dco.cholesky_block_cyclic(A, nb, rank, world, dco.torch_bcast, group=group, tail_from=tail_from, tail_chunk=chunk,
                                           tail_alone=alone)
            assert np.allclose(Lt, ref, rtol=1e-11, atol=1e-11), (group, tail_from, alone, np.abs(Lt - ref).max())
    tdist.barrier()
    tdist.destroy_process_group()
    sys.stdout.write("rank " + str(rank) + " ok" + chr(10))  # one write(): atomic on the shared pipe
    sys.stdout.flush()
    """
)


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_gloo_block_cyclic(tmp_path, world):
    """world 2: the contract's N > 1 case; world 3: the smallest job in which the tail's owner can keep out of the cyclic part."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    for attempt in range(3):  # the free-port probe can race with another process: retry on a new port
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr",
               "127.0.0.1", "--master-port", str(port), str(script)]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert all("rank %d ok" % r in out.stdout for r in range(world))


def test_socket_bootstrap_two_processes(tmp_path):
    """The default unique-id ferry (plain TCP, no torch): two real processes."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = textwrap.dedent(
        """
        import os, sys
        sys.path.insert(0, %r)
        from gptest_amd import dist
        rank = int(os.environ["RANK"])
        got = dist.socket_bootstrap(bytes(range(128)) if rank == 0 else None)
        assert got == bytes(range(128))
        sys.stdout.write("ok " + str(rank) + chr(10))
        """
    ) % ROOT
    procs = []
    for r in (0, 1):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, o + e


def test_socket_bootstrap_skips_a_foreign_listener(tmp_path):
    """The first candidate port is taken by some other service (it answers, but not with the magic tag): rank 0 moves
    to the next port of the fixed sequence and the other ranks find it there.  Three ranks."""
    import threading

    probe = socket.socket()
    probe.bind(("127.0.0.1", 0))
    port = probe.getsockname()[1]
    probe.close()
    os.environ["MASTER_PORT"] = str(port)
    try:
        ports = dist._bootstrap_ports()
    finally:
        del os.environ["MASTER_PORT"]
    foreign = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    foreign.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        foreign.bind(("127.0.0.1", ports[0]))
    except OSError:
        pytest.skip("candidate port already in use on this machine")
    foreign.listen(8)
    foreign.settimeout(0.2)
    stop = []

    def serve():
        while not stop:
            try:
                c, _ = foreign.accept()
            except OSError:
                continue
            c.sendall(b"HTTP/1.1 400 Bad Request\r\n\r\n")
            c.close()

    th = threading.Thread(target=serve, daemon=True)
    th.start()
    code = textwrap.dedent(
        """
        import os, sys
        sys.path.insert(0, %r)
        from gptest_amd import dist
        rank = int(os.environ["RANK"])
        got = dist.socket_bootstrap(bytes(range(128)) if rank == 0 else None, timeout=60.0)
        assert got == bytes(range(128))
        sys.stdout.write("ok " + str(rank) + chr(10))
        """
    ) % ROOT
    procs = []
    for r in (0, 1, 2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    stop.append(1)
    foreign.close()
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, o + e
