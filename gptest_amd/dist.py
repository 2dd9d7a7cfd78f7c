"""One process per GPU: rank discovery and RCCL communicator bring-up.

libgpk.so owns the RCCL communicator (panel broadcasts happen inside the factorisation, in C++);
the host only has to hand every rank the same 128-byte ncclUniqueId.  ``bootstrap`` is any
callable ``bytes|None -> bytes`` that broadcasts rank 0's bytes to all ranks: ``socket_bootstrap`` (default,
plain TCP next to the launcher's MASTER_PORT, no torch anywhere) or ``torch_bootstrap`` (an existing
torch.distributed process group).  Barriers and max-reductions go through RCCL itself.
"""
import ctypes
import os

from . import _lib

UID_BYTES = 128


def env_rank():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not (0 <= rank < world):
        raise ValueError("bad RANK/WORLD_SIZE: %d/%d" % (rank, world))
    return rank, local, world


def block_cyclic_owner(block, world, tail_from=None, tail_alone=False):
    """Owner rank of ownership block ``block`` (1-D block-cyclic).  Blocks from ``tail_from`` on form the
    tail block, which belongs as a whole to one rank (gpk_model::ct in gptest_amd/csrc/gpk_core.h: the chain-bound last
    columns are factored by one rank's tile pipeline): the cyclic owner of block ``tail_from``, or -- ``tail_alone``,
    gpk_model::Pc -- the last rank, which then keeps out of the cyclic part (the others share it, cyclic over world - 1)."""
    if tail_from is not None and tail_alone and world >= 3:
        return world - 1 if block >= tail_from else block % (world - 1)
    if tail_from is not None and block >= tail_from:
        block = tail_from
    return block % world


def local_blocks(rank, world, nblocks, tail_from=None, tail_alone=False):
    """Global block-column indices owned by ``rank``."""
    return [b for b in range(nblocks) if block_cyclic_owner(b, world, tail_from, tail_alone) == rank]


def torch_bootstrap(payload):
    import torch
    import torch.distributed as dist

    t = torch.zeros(UID_BYTES, dtype=torch.uint8)
    if payload is not None:
        t = torch.tensor(list(payload), dtype=torch.uint8)
    dist.broadcast(t, src=0)
    return bytes(t.tolist())


_MAGIC = b"GPKUID1\n"
_PORT_TRIES = 8


def _bootstrap_ports():
    base = (int(os.environ.get("MASTER_PORT", "29500")) + 7919) % 65536
    if base < 1024:
        base += 20000
    return [1024 + (base - 1024 + 101 * i) % (65536 - 1024) for i in range(_PORT_TRIES)]


def socket_bootstrap(payload, rank=None, world=None, timeout=120.0):
    """Broadcast rank 0's bytes over plain TCP (no torch in the product path).  Rank 0 listens next to the
    launcher's rendezvous port (MASTER_PORT + 7919, or the next of a fixed sequence of ports if that one is taken),
    the other ranks connect with retries; every message starts with a magic tag so that a foreign listener on one of
    the candidate ports is recognised and skipped."""
    import socket
    import time

    if rank is None or world is None:
        rank, _, world = env_rank()
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    ports = _bootstrap_ports()
    if rank == 0:
        srv = None
        for port in ports:
            try:
                cand = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                cand.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                cand.bind((host, port))
                srv = cand
                break
            except OSError:
                cand.close()
        if srv is None:
            raise OSError("no free bootstrap port among %r on %s" % (ports, host))
        srv.listen(world)
        srv.settimeout(timeout)
        try:
            for _ in range(world - 1):
                c, _addr = srv.accept()
                c.sendall(_MAGIC + payload)
                c.close()
        finally:
            srv.close()
        return payload
    deadline = time.time() + timeout
    want = len(_MAGIC) + UID_BYTES
    while True:
        for port in ports:
            try:
                c = socket.create_connection((host, port), timeout=2.0)
            except OSError:
                continue
            try:
                c.settimeout(10.0)
                buf = b""
                while len(buf) < want:
                    chunk = c.recv(want - len(buf))
                    if not chunk:
                        break
                    buf += chunk
            except OSError:
                buf = b""
            finally:
                c.close()
            if len(buf) == want and buf.startswith(_MAGIC):
                return buf[len(_MAGIC):]
        if time.time() > deadline:
            raise TimeoutError("rank %d: no bootstrap server answered on %s ports %r" % (rank, host, ports))
        time.sleep(0.05)


def init_comm(rank, world, bootstrap=socket_bootstrap, device=None):
    """Bind to the GPU, create the RCCL communicator inside libgpk.so."""
    _lib.init(device)
    l = _lib.lib()
    if world == 1:
        _lib.check(l.gpk_comm_init(0, 1, None), "gpk_comm_init")
        return
    buf = ctypes.create_string_buffer(UID_BYTES)
    payload = None
    if rank == 0:
        _lib.check(l.gpk_comm_unique_id(buf), "gpk_comm_unique_id")
        payload = buf.raw
    uid = bootstrap(payload)
    if len(uid) != UID_BYTES:
        raise ValueError("bootstrap returned %d bytes, expected %d" % (len(uid), UID_BYTES))
    ubuf = ctypes.create_string_buffer(uid, UID_BYTES)
    _lib.check(l.gpk_comm_init(rank, world, ubuf), "gpk_comm_init")
