"""NumPy restatement of the multi-GPU algorithm (TEST INFRASTRUCTURE ONLY): right-looking blocked
Cholesky with 1-D block-cyclic column ownership and one broadcast of the factored panel per step
(SURVEY.md 8e).  The reference has no distributed code; this exists so the N>1 schedule can be checked on
CPU over gloo (tests/test_dist_gloo.py).  Each rank holds only its own block columns of A plus the
panels it received, exactly like gptest_amd/csrc/gpk_core.hip: potrf_group; `group` consecutive panels form one
ownership block (gpk_model::OB, option dist_group): panel g belongs to rank (g // group) % world.  With `tail_from`
(a panel index on an ownership-block boundary) the panels from there on form the tail block of gpk_model::ct: one rank owns
and factors all of them, the others receive them `tail_chunk` panels per broadcast (dist_tail_phase); with `tail_alone` that rank
is the last one and owns nothing else (gpk_model::Pc: the cyclic part is shared by the other world - 1 ranks)."""
import numpy as np


def torch_bcast(arr, root):
    import torch
    import torch.distributed as tdist

    t = torch.from_numpy(arr)
    tdist.broadcast(t, src=root)
    return t.numpy()


def cholesky_block_cyclic(A, nb, rank, world, bcast, group=1, tail_from=None, tail_chunk=1, tail_alone=False):
    N = A.shape[0]
    nP = (N + nb - 1) // nb
    if tail_from is not None:
        assert 0 < tail_from < nP and tail_from % group == 0

    def owner(g):
        if tail_from is not None and tail_alone and world >= 3:
            return world - 1 if g >= tail_from else (g // group) % (world - 1)
        if tail_from is not None and g >= tail_from:
            g = tail_from
        return (g // group) % world

    mine = [g for g in range(nP) if owner(g) == rank]
    local = {g: A[:, g * nb:min(N, (g + 1) * nb)].copy() for g in mine}  # my block columns (all rows)
    L = np.zeros_like(A)

    def factor(p):
        c0, c1 = p * nb, min(N, (p + 1) * nb)
        w = c1 - c0
        panel = np.empty((N - c0, w))
        blk = local[p]
        L11 = np.linalg.cholesky(blk[c0:c1, :])
        panel[:w] = L11
        if c1 < N:  # L21 = A21 * inv(L11)^T, the inverse applied as a product like the GPU path
            panel[w:] = blk[c1:, :] @ np.linalg.inv(L11).T
        return panel

    def update(p, panel):
        c0, c1 = p * nb, min(N, (p + 1) * nb)
        P21 = panel[c1 - c0:]
        for g in mine:
            if g <= p:
                continue
            g0, g1 = g * nb, min(N, (g + 1) * nb)
            # rows below panel p's diagonal block; B rows = the panel rows of MY global columns
            local[g][c1:, :] -= P21 @ P21[g0 - c1:g1 - c1].T

    p = 0
    while p < nP:
        if tail_from is not None and p >= tail_from:
            # the tail block: its owner factors a chunk of panels without talking to anybody, then the chunk travels as one message
            q1 = min(nP, p + tail_chunk)
            rows = N - p * nb
            chunk = np.zeros((rows, min(N, q1 * nb) - p * nb))
            if owner(p) == rank:
                for q in range(p, q1):
                    panel = factor(q)
                    chunk[q * nb - p * nb:, q * nb - p * nb:q * nb - p * nb + panel.shape[1]] = panel
                    update(q, panel)
            chunk = bcast(chunk, owner(p))
            L[p * nb:, p * nb:p * nb + chunk.shape[1]] = chunk
            p = q1
            continue
        c0, c1 = p * nb, min(N, (p + 1) * nb)
        panel = factor(p) if owner(p) == rank else np.empty((N - c0, c1 - c0))
        panel = bcast(panel, owner(p))
        L[c0:, c0:c1] = panel
        update(p, panel)
        p += 1
    return np.tril(L)
