"""One process per GPU: rank discovery and RCCL communicator bring-up.

libgpk.so owns the RCCL communicator (panel broadcasts happen inside the factorisation, in C++);
the host only has to hand every rank the same 128-byte ncclUniqueId.  ``bootstrap`` is any
callable ``bytes|None -> bytes`` that broadcasts rank 0's bytes to all ranks; ``torch_bootstrap``
does it over a torch.distributed (gloo) process group -- torch is plumbing here, never compute.
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


def block_cyclic_owner(block, world):
    """Owner rank of outer block column ``block`` (1-D block-cyclic)."""
    return block % world


def local_blocks(rank, world, nblocks):
    """Global block-column indices owned by ``rank``."""
    return list(range(rank, nblocks, world))


def torch_bootstrap(payload):
    import torch
    import torch.distributed as dist

    t = torch.zeros(UID_BYTES, dtype=torch.uint8)
    if payload is not None:
        t = torch.tensor(list(payload), dtype=torch.uint8)
    dist.broadcast(t, src=0)
    return bytes(t.tolist())


def init_comm(rank, world, bootstrap=torch_bootstrap, device=None):
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
