"""One process per GPU: row-block partition of a matrix and the allgatherv of the operand vector.

The reference has no distributed code at all; this is the MI355X-side design for
the one exchange step the mxv / vxm path has: GPU p owns a contiguous block of
output rows (block boundaries balance *entries*, not rows — R-MAT is skewed) and needs all of `u`,
so before each product every rank sends its slice of the vector to every other rank directly
(grouped point-to-point over xGMI via RCCL: fully connected, no ring), straight into the HBM
buffer the SpMV kernel gathers from.  torch.distributed is plumbing only.
"""
import numpy as np


def balanced_row_blocks(row_weight_prefix, nparts):
    """Boundaries b[0..nparts] of contiguous row blocks with ~equal total weight.

    `row_weight_prefix[r]` = total weight (entries) of rows < r, length nrows + 1, non-decreasing.
    """
    prefix = np.asarray(row_weight_prefix, dtype=np.int64)
    nrows = len(prefix) - 1
    total = int(prefix[-1])
    bounds = [0]
    for p in range(1, nparts):
        target = (total * p) // nparts
        b = int(np.searchsorted(prefix, target, side="left"))
        bounds.append(min(max(b, bounds[-1]), nrows))
    bounds.append(nrows)
    return bounds


def rmat_expected_row_prefix(scale, nparts_hint=0):
    """Analytic expected entry prefix of an R-MAT row block: lets every rank agree on balanced
    boundaries without generating the whole graph.  Row i has expected share
    prod_l (0.76 if bit l of i is 0 else 0.24)."""
    n = 1 << scale
    w = np.ones(1, np.float64)
    for _ in range(scale):                       # most significant bit first
        w = np.concatenate([w * 0.76, w * 0.24])
    prefix = np.zeros(n + 1, np.float64)
    np.cumsum(w, out=prefix[1:])
    return (prefix * (1 << 40)).astype(np.int64)


def allgatherv_into(full, mine, bounds, rank, world, dist, mode=None):
    """full[bounds[p]:bounds[p+1]] <- rank p's `mine` for every p (tensors on this rank's device).

    mode "p2p" (default): grouped isend/irecv — every pair of ranks exchanges directly (one xGMI hop on MI355X).
    mode "broadcast": one broadcast per rank — slower, but available on every backend (used for the gloo smoke runs)."""
    import os
    mode = mode or os.environ.get("GRB_DIST_ALLGATHER", "p2p")
    full[bounds[rank]:bounds[rank + 1]].copy_(mine)
    if world == 1:
        return
    if mode == "broadcast":
        for p in range(world):
            dist.broadcast(full[bounds[p]:bounds[p + 1]], src=p)
        return
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        ops.append(dist.P2POp(dist.isend, mine, peer))
        ops.append(dist.P2POp(dist.irecv, full[bounds[peer]:bounds[peer + 1]], peer))
    for req in dist.batch_isend_irecv(ops):
        req.wait()


class DeviceArray:
    """A raw HBM address seen through __cuda_array_interface__ so torch can wrap it without a copy."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def as_torch(ptr, n, typestr, device):
    import torch
    return torch.as_tensor(DeviceArray(ptr, n, typestr), device=device)
