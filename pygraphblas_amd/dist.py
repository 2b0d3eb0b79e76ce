"""One process per GPU: the row-partitioned forms of the hot path's three workloads and their exchange steps.

The reference has no distributed code at all; this is the MI355X-side design for the
exchange steps BASELINE.json's north star names: GPU p owns a contiguous block of output
rows (block boundaries balance *work* — entries, or the flop bound of a masked product — not rows: R-MAT is
skewed) and needs all of the operand vector, so before each product every rank sends its slice to every
other rank directly (grouped point-to-point over xGMI: fully connected, no ring), straight into the HBM
buffer the SpMV kernel gathers from.

The exchange itself lives in the library (`GrBX_dist_*`, pygraphblas_amd/csrc/grb_dist.cpp: RCCL linked
directly, its own HIP stream so that the diagonal block of the product overlaps it) — `Comm("rccl")` binds it.
`Comm("host")` moves the same slices through torch.distributed with host staging (any backend, gloo in the
tests): it exists so that the N>1 code paths can run on a machine with one GPU or none, where RCCL refuses
to put two ranks on one device.

  pagerank(...)        gap/prmark.py:8-30, FP32 PLUS_SECOND, row blocks of A' split into a diagonal and an
                       off-diagonal part, allgatherv of w overlapped with the diagonal product, all-reduce of rdiff
  bfs_levels(...)      demo/Introduction-to-GraphBLAS-with-Python.ipynb:4301-4313, the frontier gathered as bits
  triangle_count(...)  demo/TriangleCentrality.ipynb:1446-1449, L replicated, row blocks balanced by flop bound,
                       all-reduce of the INT64 count
"""
import ctypes as C

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


# ---- the communicator ------------------------------------------------------------------------------------------------
class Comm:
    """rank / world + the exchange primitives.  transport "rccl": the library's RCCL communicator; `share(id_bytes)`
    must return rank 0's 128 bytes on every rank (a torch.distributed / MPI / file broadcast — any channel).
    transport "host": torch.distributed collectives on host copies (tests; one GPU or none)."""

    def __init__(self, rank=0, world=1, transport="rccl", share=None, tdist=None, device=None):
        from . import lib
        self.rank, self.world, self.transport, self.lib = rank, world, transport, lib
        self.tdist, self.device = tdist, device
        self._pending = None
        if world > 1 and transport == "rccl":
            buf = C.create_string_buffer(128)
            ident, err = b"", None
            if rank == 0:
                try:
                    _check(lib.GrBX_dist_unique_id(buf, C.c_int(128)))
                    ident = bytes(buf.raw)
                except Exception as e:          # noqa: BLE001 - the other ranks wait in share(): tell them instead of leaving them there
                    err = e
            ident = share(ident)
            if not ident:
                raise err or RuntimeError("rank 0 could not create the RCCL communicator id")
            _check(lib.GrBX_dist_init(C.c_int(rank), C.c_int(world), C.c_char_p(ident), C.c_int(128)))
        elif world > 1 and transport != "host":
            raise ValueError("transport must be 'rccl' or 'host'")

    def close(self):
        if self.world > 1 and self.transport == "rccl":
            self.lib.GrBX_dist_finalize()

    @staticmethod
    def _bounds(bounds):
        return np.ascontiguousarray(bounds, np.uint64)

    # full[bounds[p]:bounds[p+1]] <- rank p's local vector, for every p
    def allgatherv_start(self, full, local, bounds, presence=False):
        b = self._bounds(bounds)
        if self.world == 1 or self.transport == "rccl":
            _check(self.lib.GrBX_Vector_allgatherv_start(full._h, local._h, b.ctypes.data_as(C.c_void_p), C.c_int(1 if presence else 0)), full)
        else:
            self._host_gather(full, local, bounds, presence)

    def wait(self):
        if self.world == 1 or self.transport == "rccl":
            _check(self.lib.GrBX_dist_wait())

    def allgatherv(self, full, local, bounds, presence=False):
        self.allgatherv_start(full, local, bounds, presence)
        self.wait()

    def allgatherv_bits(self, full, local, bounds):
        b = self._bounds(bounds)
        if self.world == 1 or self.transport == "rccl":
            _check(self.lib.GrBX_Vector_allgatherv_bits(full._h, local._h, b.ctypes.data_as(C.c_void_p)), full)
        else:
            self._host_gather(full, local, bounds, True, bits=True)

    def allreduce(self, value, typ, op="PLUS"):
        """A host scalar reduced over the ranks with the type's monoid operator `op`."""
        if self.world == 1:
            return value
        if self.transport == "rccl":
            x = np.array([value], typ._np)
            _check(self.lib.GrBX_dist_allreduce(x.ctypes.data_as(C.c_void_p), C.c_uint64(1), C.c_void_p(typ._h), C.c_void_p(getattr(typ, op).get_op())))
            return x[0].item()
        import torch
        t = torch.tensor([value], dtype=torch.float64 if typ._np in (np.float32, np.float64) else torch.int64)
        self.tdist.all_reduce(t, op={"PLUS": self.tdist.ReduceOp.SUM, "MIN": self.tdist.ReduceOp.MIN, "MAX": self.tdist.ReduceOp.MAX}[op])
        return typ._np(t[0].item()).item()

    def _host_gather(self, full, local, bounds, presence, bits=False):
        vals, pres = local.to_dense_arrays()
        if bits:
            vals = ((vals != 0) & (pres != 0)).astype(full.type._np); pres = (vals != 0).astype(np.uint8)
        parts = [None] * self.world
        self.tdist.all_gather_object(parts, (vals, pres))
        allv = np.concatenate([p[0] for p in parts]); allp = np.concatenate([p[1] for p in parts])
        assert len(allv) == bounds[-1]
        fv, fp, _ = full.device_view()
        _upload(fv, allv); _upload(fp, allp if presence else np.ones(len(allp), np.uint8))
        _check(self.lib.GrBX_Vector_device_touch(full._h), full)


def bound_transport():
    """File name of the RCCL-ABI library the exchange is bound to ("" while nothing is bound): librccl on a multi-GPU node, or the
    test stand-in GRB_MI355X_RCCL names (tests/libfake_rccl.so: two ranks on one GPU, hipIpc copies)."""
    from . import lib
    buf = C.create_string_buffer(512)
    lib.GrBX_dist_transport(buf, C.c_int(512))
    return buf.value.decode()


def _check(info, obj=None):
    from .base import check
    check(info, obj)


def _upload(dev_ptr, host_array):
    """host numpy -> HBM at a raw address (hipMemcpy through the HIP runtime the library already loaded)."""
    hip = C.CDLL("libamdhip64.so")
    a = np.ascontiguousarray(host_array)
    rc = hip.hipMemcpy(C.c_void_p(dev_ptr), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), C.c_int(1))
    if rc != 0:
        raise RuntimeError(f"hipMemcpy failed: {rc}")


# ---- partitioning helpers (torch tensors in HBM; workload plumbing, like rmat.py) -----------------------------------------
def split_csr_columns(rowptr, col, c0, c1, vals=None):
    """Split a CSR row block (torch tensors, int32 bit patterns of u32) into the entries whose column lies in [c0, c1)
    and the rest.  Returns ((rowptr_d, col_d, vals_d), (rowptr_o, col_o, vals_o))."""
    import torch
    nr = rowptr.numel() - 1
    rp = rowptr.to(torch.int64) & 0xFFFFFFFF
    cl = col.to(torch.int64) & 0xFFFFFFFF
    rows = torch.repeat_interleave(torch.arange(nr, device=col.device), rp[1:] - rp[:-1])
    inside = (cl >= c0) & (cl < c1)
    out = []
    for m in (inside, ~inside):
        cnt = torch.bincount(rows[m], minlength=nr)
        p = torch.zeros(nr + 1, dtype=torch.int64, device=col.device); p[1:] = torch.cumsum(cnt, 0)
        out.append((p.to(torch.int32), col[m].contiguous(), vals[m].contiguous() if vals is not None else None))
    return out[0], out[1]


def flop_balanced_row_blocks(rowptr, col, nparts):
    """Row blocks of a masked product C<M> = A*B with A = M = B = the given CSR, balanced by the flop bound
    sum_{k in A(i,:)} nnz(B(k,:)) of every row."""
    import torch
    rp = rowptr.to(torch.int64) & 0xFFFFFFFF
    deg = rp[1:] - rp[:-1]
    cl = col.to(torch.int64) & 0xFFFFFFFF
    w = torch.zeros(cl.numel() + 1, dtype=torch.int64, device=col.device)
    w[1:] = torch.cumsum(deg[cl], 0)                         # flop bound of the entries before entry e
    rowflops_prefix = w[rp]                                  # ... of the rows before row r
    return balanced_row_blocks(rowflops_prefix.cpu().numpy(), nparts)


# ---- the three workloads, row-partitioned --------------------------------------------------------------------------------
def pagerank(comm, Dm, Om, d, n, bounds, damping=0.85, itermax=100, tol=1e-4, fixed_iterations=None):
    """gap/prmark.py:8-30 on a row block: this rank owns vertices [bounds[rank], bounds[rank+1]).
    Dm / Om: the rows of A' (the transpose of the adjacency matrix) of those vertices, split into the columns this rank
    owns (diagonal block) and the others; both (r1-r0) x n, FP32 or a BOOL pattern.  d: out-degrees of the owned vertices
    (FP32 vector of length r1-r0, no entry for dangling vertices — the reference's `A.reduce_vector()`).
    Per iteration: w = t / d on the slice; the allgatherv of w starts; r = teleport + Dm (+).second w runs on the local
    columns while the remote slices arrive; then r += Om (+).second w; |t - r| is summed locally and all-reduced.
    Returns (r_local, iterations, rdiff)."""
    from . import Vector, FP32
    rank = comm.rank
    nb = bounds[rank + 1] - bounds[rank]
    r = Vector.sparse(FP32, nb)
    t = Vector.sparse(FP32, nb)
    w_full = Vector.dense(FP32, n, fill=0.0)
    d.assign_scalar(damping, accum=FP32.DIV)
    r[:] = 1.0 / n
    teleport = (1 - damping) / n
    rdiff, its = 1.0, 0
    for i in range(fixed_iterations if fixed_iterations is not None else itermax):
        t, r = r, t
        w = t / d
        comm.allgatherv_start(w_full, w, bounds, presence=True)
        r[:] = teleport
        Dm.mxv(w_full, out=r, accum=FP32.PLUS, semiring=FP32.PLUS_SECOND)
        comm.wait()
        Om.mxv(w_full, out=r, accum=FP32.PLUS, semiring=FP32.PLUS_SECOND)
        t -= r
        t.apply(FP32.ABS, out=t)
        rdiff = comm.allreduce(t.reduce_float(), FP32, "PLUS")
        its = i + 1
        if fixed_iterations is None and rdiff <= tol:
            break
    return r, its, rdiff


def bfs_levels(comm, Arows, n, bounds, source):
    """The reference's BFS loop on a row block.  Arows: rows [r0, r1) of A' (= of A for an undirected graph), (r1-r0) x n
    BOOL.  The visited/level vector is kept both whole (the operand of the product) and as this rank's slice (its mask);
    each level's new frontier is gathered as one bit per vertex.  Returns (levels of the owned vertices, depth)."""
    from . import Vector, UINT8, BOOL, descriptor as D
    rank = comm.rank
    r0, r1 = bounds[rank], bounds[rank + 1]
    v_full = Vector.sparse(UINT8, n); v_loc = Vector.sparse(UINT8, r1 - r0)
    q_full = Vector.sparse(BOOL, n); q_loc = Vector.sparse(BOOL, r1 - r0)
    q_full[source] = True
    if r0 <= source < r1:
        q_loc[source - r0] = True
    level = 1
    while q_full.reduce_bool() and level <= n:
        v_full.assign_scalar(level, mask=q_full)
        v_loc.assign_scalar(level, mask=q_loc)
        Arows.mxv(v_full, mask=v_loc, out=q_loc, semiring=BOOL.LOR_LAND, desc=D.RC)      # = v.vxm(A, mask=v, out=q, desc=RC) on the owned columns
        comm.allgatherv_bits(q_full, q_loc, bounds)
        level += 1
    return v_loc, level - 1


def triangle_count(comm, Lrows, L):
    """L.mxm(L, PLUS_PAIR, mask=L).reduce_int() with the rows of A and of the mask partitioned: Lrows = this rank's rows of
    L ((r1-r0) x n), L replicated.  The per-rank counts are all-reduced (INT64, exact)."""
    from . import INT64
    local = Lrows.mxm(L, semiring=INT64.PLUS_PAIR, mask=Lrows).reduce_int()
    return comm.allreduce(local, INT64, "PLUS")
