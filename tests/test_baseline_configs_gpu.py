"""BASELINE.json configs[0..3] (+ the single-GPU step of configs[4]) at their STATED sizes, through the C ABI on the MI355X,
against the oracle (oracle/grb_oracle.c) — and scipy as a second opinion where it applies.

  configs[0]  1000x1000 `Matrix.random(FP64, 10 000, seed=42)` mxv PLUS_TIMES_FP64   (reference generator: pygraphblas/matrix.py:499-571)
  configs[1]  R-MAT scale-22 FP64 PLUS_TIMES SpMV                                     (also what bench.py times)
  configs[2]  R-MAT scale-22 BOOL LOR_LAND BFS, the reference's loop                  (demo/Introduction-to-GraphBLAS-with-Python.ipynb:4301-4313)
  configs[3]  triangle count L.mxm(L, PLUS_PAIR, mask=L).reduce_int() on R-MAT-22     (demo/TriangleCentrality.ipynb:1446-1449)
  configs[4]  one FP32 PageRank step (PLUS_SECOND, T0, accum PLUS) on R-MAT-22        (gap/prmark.py:17-29); the BOOL-pattern
              matrix of the reference's driver (gap/prmark.py:47 `.pattern()`) with an FP32 semiring at nnz >= 2^22;
              and the step + the whole loop at the STATED scale-25 on one GPU
  MIN_PLUS    the reference's shortest-path loop (repeated vxm, accum MIN) on R-MAT-22, INT64 and FP64 weights
Tolerances: bit-exact for BOOL / integer results, 1e-6 relative for FP64 / FP32 (BASELINE.json north_star).
The scale-22 graphs are generated in HBM by pygraphblas_amd.rmat (counter-based R-MAT) and the oracle's typed
OpenMP loops run on the same CSR arrays; every case finishes in seconds.
"""
import numpy as np
import pytest

from oracle import oracle as O
from helpers import TYPE

pytestmark = pytest.mark.gpu

SCALE = 22


@pytest.fixture(scope="module")
def torch_dev(gpu):
    import torch
    return torch, torch.device("cuda", 0)


# ---- configs[0] ---------------------------------------------------------------------------------------------------------
def test_config0_reference_random_1000_mxv(gb, gpu):
    """`Matrix.random` reproduces the reference's draw sequence (tests/test_matrix.py:1060-1064), then the 1000x1000 FP64 mxv."""
    import random
    v = gb.Matrix.random(gb.INT8, 4, 10, 10, seed=42)
    assert len(v) == 4
    assert v.to_scipy_sparse().data.tolist() == [62, 46, -70, 24]
    A = gb.Matrix.random(gb.FP64, 10_000, 1000, 1000, seed=42)
    assert 9_000 < A.nvals <= 10_000                               # repeated coordinates overwrite
    x = np.array([random.random() for _ in range(1000)])           # continues the same Python stream, as a caller of the reference would
    u = gb.Vector.from_arrays(np.arange(1000, dtype=np.uint64), x, 1000, gb.FP64)
    w = A.mxv(u, semiring=gb.FP64.PLUS_TIMES)
    gi, gx = w.to_arrays()
    I, J, X = A.to_arrays()
    exp = O.mxv(O.col_vector("FP64", 1000), O.Tuples("FP64", 1000, 1000, I, J, X), O.col_vector("FP64", 1000, np.arange(1000), x), "PLUS", "TIMES", "FP64")
    assert np.array_equal(gi, exp.I)
    assert np.allclose(gx, exp.X, rtol=1e-6, atol=0.0)
    S = A.to_scipy_sparse().tocsr()
    y = S @ x
    rows = np.flatnonzero(np.diff(S.indptr))
    assert np.array_equal(gi.astype(np.int64), rows)
    assert np.allclose(gx, y[rows], rtol=1e-6, atol=0.0)


# ---- configs[1] ---------------------------------------------------------------------------------------------------------
def test_config1_rmat22_fp64_spmv(gb, torch_dev):
    torch, dev = torch_dev
    from pygraphblas_amd import rmat
    n = 1 << SCALE
    rowptr, col = rmat.csr_torch(SCALE, dev, seed=42)
    nnz = int(col.numel())
    vals = rmat.values_torch(nnz, dev, seed=43)
    xs = rmat.values_torch(n, dev, seed=44)
    A = gb.Matrix.from_csr(gb.FP64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    x = gb.Vector.from_dense_array((xs.data_ptr(), n), gb.FP64, device=True)
    y, pres = O.fast_spmv(rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32), vals.cpu().numpy(), xs.cpu().numpy())
    # the FIRST product of a matrix runs kernel W (a sub-millisecond plan: a caller that multiplies once must not wait for a panel copy) ...
    w0 = A.mxv(x, semiring=gb.FP64.PLUS_TIMES)
    assert "k_spmv_wavepipe" in gb.last_kernel_plan(), gb.last_kernel_plan()
    g0, p0 = w0.to_dense_arrays()
    assert np.array_equal(p0 != 0, pres != 0) and np.allclose(g0[pres != 0], y[pres != 0], rtol=1e-6, atol=0.0)
    # ... the second builds kernel X's plan: the north-star kernel
    w = A.mxv(x, semiring=gb.FP64.PLUS_TIMES)
    assert "k_spmv_xcd" in gb.last_kernel_plan(), gb.last_kernel_plan()
    gy, gp = w.to_dense_arrays()
    assert np.array_equal(gp != 0, pres != 0)
    assert np.allclose(gy[pres != 0], y[pres != 0], rtol=1e-6, atol=0.0)
    # a third call reuses the plan and must give the same bits (fixed summation order)
    w2 = A.mxv(x, semiring=gb.FP64.PLUS_TIMES)
    gy2, _ = w2.to_dense_arrays()
    assert np.array_equal(gy2[pres != 0], gy[pres != 0])


def test_config1_on_the_surveyed_pcg64_graph(gb, torch_dev, capsys):
    """configs[1] on the graph of SURVEY.md §8d to the letter — `numpy.random.Generator(PCG64(42))`, one draw per bit level, values from PCG64(43), the
    operand from PCG64(44) (pygraphblas_amd.rmat.csr_numpy_pcg64) — instead of the counter-based stream every other number of this repository is on:
    parity against the oracle (pattern exact, values 1e-6) and the kernel-time fraction of the 8 TB/s roofline, printed and required to sit within 0.03
    of what the counter-based graph gives on the same box (the headline does not depend on the generator)."""
    import ctypes as C
    torch, dev = torch_dev
    from pygraphblas_amd import rmat
    n = 1 << SCALE

    def frac_of(A, x, nnz):
        w = A.mxv(x, semiring=gb.FP64.PLUS_TIMES); w = A.mxv(x, semiring=gb.FP64.PLUS_TIMES, out=w)        # kernel W, then kernel X's plan
        assert "k_spmv_xcd" in gb.last_kernel_plan(), gb.last_kernel_plan()
        for _ in range(5):
            A.mxv(x, semiring=gb.FP64.PLUS_TIMES, out=w)
        best = 1e9
        for _ in range(5):
            gb.lib.GrBX_timer_start()
            for _ in range(20):
                A.mxv(x, semiring=gb.FP64.PLUS_TIMES, out=w)
            ms = C.c_float(0); gb.lib.GrBX_timer_stop(C.byref(ms)); best = min(best, ms.value / 20)
        alg = nnz * 12 + (n + 1) * 4 + 2 * 8 * n
        return w, alg / (best * 1e-3) / 8e12, best
    rp, col = rmat.csr_numpy_pcg64(SCALE, seed=42)
    nnz = len(col)
    assert 6.4e7 < nnz < 6.72e7 and rp[-1] == nnz                  # 16 * 2^22 sampled edges less the duplicates
    vals = rmat.values_numpy_pcg64(nnz, 43); xs = rmat.values_numpy_pcg64(n, 44)
    A = gb.Matrix.from_csr(gb.FP64, n, n, rp, col, vals)
    x = gb.Vector.from_dense_array(xs, gb.FP64)
    w, frac, ms = frac_of(A, x, nnz)
    y, pres = O.fast_spmv(rp, col, vals, xs)
    gy, gp = w.to_dense_arrays()
    assert np.array_equal(gp != 0, pres != 0) and np.allclose(gy[pres != 0], y[pres != 0], rtol=1e-6, atol=0.0)
    del A, x, w
    rowptr, ccol = rmat.csr_torch(SCALE, dev, seed=42)
    cn = int(ccol.numel()); cv = rmat.values_torch(cn, dev, seed=43); cx = rmat.values_torch(n, dev, seed=44)
    B = gb.Matrix.from_csr(gb.FP64, n, n, rowptr.data_ptr(), ccol.data_ptr(), (cv.data_ptr(), cn), device=True)
    xb = gb.Vector.from_dense_array((cx.data_ptr(), n), gb.FP64, device=True)
    _, frac_c, ms_c = frac_of(B, xb, cn)
    with capsys.disabled():
        print(f"\n[configs[1] on the PCG64(42) graph of SURVEY 8d: nnz {nnz}, {ms:.4f} ms per product, {frac:.4f} of 8 TB/s;  counter-based graph: nnz {cn}, {ms_c:.4f} ms, {frac_c:.4f}]")
    assert frac >= 0.40 and abs(frac - frac_c) <= 0.03, (frac, frac_c)


@pytest.mark.parametrize("typ", ["INT64", "FP32", "INT32"])
def test_rmat22_plus_times_spmv_other_types_exact(gb, torch_dev, typ):
    """configs[1]'s product in the non-headline types, at the stated size.  Small integer values keep every row sum exactly
    representable (hub rows: 10^5 terms of at most 21 < 2^24), so the panel pipeline's INT64 / INT32 / FP32 instantiations must
    reproduce the oracle's FP64 sums of the same integers bit for bit, whatever order they add in."""
    torch, dev = torch_dev
    from pygraphblas_amd import rmat
    n = 1 << SCALE
    rowptr, col = rmat.csr_torch(SCALE, dev, seed=42)
    nnz = int(col.numel())
    tt = {"INT64": torch.int64, "FP32": torch.float32, "INT32": torch.int32}[typ]
    vals_i = (rmat.values_torch(nnz, dev, seed=43) * 7).to(torch.int64) + 1           # 1 .. 7
    xs_i = (rmat.values_torch(n, dev, seed=44) * 4).to(torch.int64)                   # 0 .. 3
    vals, xs = vals_i.to(tt), xs_i.to(tt)
    T = getattr(gb, typ)
    A = gb.Matrix.from_csr(T, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    x = gb.Vector.from_dense_array((xs.data_ptr(), n), T, device=True)
    y, pres = O.fast_spmv(rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32), vals_i.cpu().numpy().astype(np.float64), xs_i.cpu().numpy().astype(np.float64))
    w = A.mxv(x, semiring=T.PLUS_TIMES)                            # first product of the matrix: kernel W (FP32: kernel X at once — its blocked sums are the accurate ones)
    assert ("k_spmv_xcd" if typ == "FP32" else "k_spmv_wavepipe") in gb.last_kernel_plan(), gb.last_kernel_plan()
    gy, gp = w.to_dense_arrays()
    assert np.array_equal(gp != 0, pres != 0) and np.array_equal(gy[pres != 0].astype(np.float64), y[pres != 0])
    w = A.mxv(x, semiring=T.PLUS_TIMES)                            # second: kernel X
    assert "k_spmv_xcd" in gb.last_kernel_plan()
    assert ("values=int16" in gb.last_kernel_plan()) == typ.startswith("INT"), gb.last_kernel_plan()      # (round 6: integer values that fit 16 bits are kept as int16 in the plan)
    gy, gp = w.to_dense_arrays()
    assert np.array_equal(gp != 0, pres != 0)
    assert np.array_equal(np.asarray(gy, np.float64)[pres != 0], y[pres != 0])


@pytest.mark.parametrize("typ", ["INT64", "INT32", "UINT64", "UINT32"])
def test_kernel_x_narrow_value_plane_at_its_boundaries(gb, torch_dev, typ, monkeypatch):
    """Round 6: kernel X's plan keeps the values of an integer matrix as int16 when ALL of them fit (k_spmv_tiles<..., VB = 2>: 2 instead of 4 / 8 bytes of an
    entry's stream — the weights 1 ... 255 of the shortest-path problem).  Values across the whole int16 range with both extremes present are narrowed, one value
    just beyond keeps the plan wide; PLUS_TIMES and (signed types) MIN_PLUS against numpy's exact integer arithmetic either way, and the switch that turns
    the narrow plane off gives the same results."""
    torch, dev = torch_dev
    from pygraphblas_amd import rmat
    S = 17; n = 1 << S
    rowptr, col = rmat.csr_torch(S, dev, seed=42)
    nnz = int(col.numel())
    rp = rowptr.cpu().numpy().view(np.uint32).astype(np.int64); ci = col.cpu().numpy().view(np.uint32).astype(np.int64)
    signed = typ.startswith("INT"); T = getattr(gb, typ); npt = O.NP[typ]
    rng = np.random.default_rng(5)
    xs = rng.integers(0, 4, n).astype(np.int64)
    nonempty = rp[1:] > rp[:-1]
    monkeypatch.setenv("GRB_MI355X_SPMV", "xcd")
    for case, narrow_env in (("fits", "1"), ("edge", "1"), ("beyond", "1"), ("edge", "0")):
        lo = -32768 if signed else 0
        v = rng.integers(1, 256, nnz).astype(np.int64) if case == "fits" else rng.integers(lo, 32768, nnz).astype(np.int64)
        if case != "fits": v[0] = 32767; v[1] = lo
        if case == "beyond": v[nnz // 2] = 32768; v[nnz // 3] = (-32769 if signed else 40000)
        monkeypatch.setenv("GRB_MI355X_XT_NARROW", narrow_env)
        vt = torch.from_numpy(v.astype(npt).view(np.int64 if npt().itemsize == 8 else np.int32)).to(dev)
        A = gb.Matrix.from_csr(T, n, n, rowptr.data_ptr(), col.data_ptr(), (vt.data_ptr(), nnz), device=True)
        xt = torch.from_numpy(xs.astype(npt).view(np.int64 if npt().itemsize == 8 else np.int32)).to(dev)
        x = gb.Vector.from_dense_array((xt.data_ptr(), n), T, device=True)
        prod = v * xs[ci]
        want = np.zeros(n, np.int64); np.add.at(want, np.repeat(np.arange(n), np.diff(rp)), prod)
        w = A.mxv(x, semiring=T.PLUS_TIMES)
        plan = gb.last_kernel_plan()
        assert "k_spmv_xcd" in plan and ("values=int16" in plan) == (case != "beyond" and narrow_env == "1"), (case, narrow_env, plan)
        gy, gp = w.to_dense_arrays()
        assert np.array_equal(gp != 0, nonempty), (typ, case)
        assert np.array_equal(gy[nonempty].astype(np.int64), want[nonempty].astype(npt).astype(np.int64)), (typ, case, "PLUS_TIMES")      # (wrap-around of the type included)
        if signed:
            w2 = A.mxv(x, semiring=T.MIN_PLUS)
            assert "k_spmv_xcd" in gb.last_kernel_plan()
            s2 = v + xs[ci]
            starts = rp[:-1][nonempty]
            mn = np.minimum.reduceat(s2, starts)
            gy2, gp2 = w2.to_dense_arrays()
            assert np.array_equal(gp2 != 0, nonempty) and np.array_equal(gy2[nonempty].astype(np.int64), mn), (typ, case, "MIN_PLUS")


# ---- configs[2] ---------------------------------------------------------------------------------------------------------
def _bfs(gb, A, start):
    from pygraphblas_amd import descriptor as D
    v = gb.Vector.sparse(gb.UINT8, A.nrows)
    q = gb.Vector.sparse(gb.BOOL, A.nrows)
    q[start] = True
    level = 1
    while q.reduce_bool() and level <= A.nrows:
        v.assign_scalar(level, mask=q)
        v.vxm(A, mask=v, out=q, desc=D.RC)
        level += 1
    return v, level - 1


def test_config2_rmat22_bfs_levels(gb, torch_dev):
    torch, dev = torch_dev
    from pygraphblas_amd import rmat
    n = 1 << SCALE
    rowptr, col = rmat.csr_torch(SCALE, dev, seed=42, symmetric=True, drop_self_loops=True)
    nnz = int(col.numel())
    vals = torch.ones(nnz, dtype=torch.bool, device=dev)
    A = gb.Matrix.from_csr(gb.BOOL, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    src = int(torch.argmax(rowptr[1:] - rowptr[:-1]))              # the vertex of maximum degree
    v, depth = _bfs(gb, A, src)
    lev, _ = v.to_dense_arrays()
    olev, odepth = O.fast_bfs(rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32), src)
    assert depth == odepth
    assert np.array_equal(lev, olev)                               # bit-exact level vector (unreached vertices hold 0 on both sides)
    assert int((lev > 0).sum()) > n // 4


# ---- configs[3] ---------------------------------------------------------------------------------------------------------
def test_config3_rmat22_triangle_count(gb, torch_dev):
    torch, dev = torch_dev
    from pygraphblas_amd import rmat
    n = 1 << SCALE
    rowptr, col = rmat.csr_torch(SCALE, dev, seed=42, symmetric=True, drop_self_loops=True, lower=True)
    nnz = int(col.numel())
    vals = torch.ones(nnz, dtype=torch.int64, device=dev)
    L = gb.Matrix.from_csr(gb.INT64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    tri = L.mxm(L, semiring=gb.INT64.PLUS_PAIR, mask=L).reduce_int()
    otri = O.fast_tricount(rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32))
    assert tri == otri                                              # INT64, bit-exact
    assert tri > 10**9


def test_config3_rmat22_masked_products_that_read_values(gb, torch_dev):
    """The masked product of configs[3] with multiplies that read the operands (the survivor queues then carry positions and are
    flushed per B row): per entry, L*L under PLUS_TIMES with every value 2 is four times the count PLUS_PAIR gives, with FP64
    values 0.5 a quarter of it (exact), and PLUS_FIRST / PLUS_SECOND with values 3 three times it — at the stated size, where
    every bin and the hub-row kernel are populated."""
    torch, dev = torch_dev
    from pygraphblas_amd import rmat
    n = 1 << SCALE
    rowptr, col = rmat.csr_torch(SCALE, dev, seed=42, symmetric=True, drop_self_loops=True, lower=True)
    nnz = int(col.numel())
    ones = torch.ones(nnz, dtype=torch.int64, device=dev)
    L1 = gb.Matrix.from_csr(gb.INT64, n, n, rowptr.data_ptr(), col.data_ptr(), (ones.data_ptr(), nnz), device=True)
    C = L1.mxm(L1, semiring=gb.INT64.PLUS_PAIR, mask=L1)
    ci, cj, cx = C.to_arrays()
    twos = ones * 2
    L2 = gb.Matrix.from_csr(gb.INT64, n, n, rowptr.data_ptr(), col.data_ptr(), (twos.data_ptr(), nnz), device=True)
    D2 = L2.mxm(L2, semiring=gb.INT64.PLUS_TIMES, mask=L2)
    di, dj, dx = D2.to_arrays()
    assert np.array_equal(ci, di) and np.array_equal(cj, dj) and np.array_equal(dx, 4 * cx)
    del D2, L2, di, dj, dx
    threes = ones * 3
    L3 = gb.Matrix.from_csr(gb.INT64, n, n, rowptr.data_ptr(), col.data_ptr(), (threes.data_ptr(), nnz), device=True)
    for sr in (gb.INT64.PLUS_FIRST, gb.INT64.PLUS_SECOND):
        E = L3.mxm(L3, semiring=sr, mask=L3); ei, ej, ex = E.to_arrays()
        assert np.array_equal(ci, ei) and np.array_equal(cj, ej) and np.array_equal(ex, 3 * cx)
        del E
    del L3
    halves = torch.full((nnz,), 0.5, dtype=torch.float64, device=dev)
    Lh = gb.Matrix.from_csr(gb.FP64, n, n, rowptr.data_ptr(), col.data_ptr(), (halves.data_ptr(), nnz), device=True)
    F = Lh.mxm(Lh, semiring=gb.FP64.PLUS_TIMES, mask=Lh); fi, fj, fx = F.to_arrays()
    assert np.array_equal(ci, fi) and np.array_equal(cj, fj) and np.array_equal(fx, 0.25 * cx.astype(np.float64))


def test_config3_rmat22_masked_product_entry_by_entry_against_the_oracle(gb, torch_dev):
    """configs[3]'s product C<L> = L (+).(x) L compared with the oracle ENTRY BY ENTRY at the stated size (the count test above sees one
    sum): PLUS_PAIR INT64 — pattern and every count bit-exact, so every mask bin, the survivor queues and the hub-row kernel are
    checked per entry on the real graph — and PLUS_TIMES FP64 with random values to 1e-6 (the atomic paths add in no fixed order)."""
    torch, dev = torch_dev
    from pygraphblas_amd import rmat
    n = 1 << SCALE
    rowptr, col = rmat.csr_torch(SCALE, dev, seed=42, symmetric=True, drop_self_loops=True, lower=True)
    nnz = int(col.numel())
    rp_h, col_h = rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32)
    ones = torch.ones(nnz, dtype=torch.int64, device=dev)
    L = gb.Matrix.from_csr(gb.INT64, n, n, rowptr.data_ptr(), col.data_ptr(), (ones.data_ptr(), nnz), device=True)
    C = L.mxm(L, semiring=gb.INT64.PLUS_PAIR, mask=L)
    crp, ccol, cx = C.to_csr()
    out, has = O.fast_masked_mxm(rp_h, col_h)
    keep = has != 0
    exp_rp = np.zeros(n + 1, np.int64); np.cumsum(np.add.reduceat(np.concatenate([keep.astype(np.int64), [0]]), rp_h[:-1].astype(np.int64)) * (np.diff(rp_h.astype(np.int64)) > 0), out=exp_rp[1:])
    assert np.array_equal(crp.astype(np.int64), exp_rp)                        # the pattern: same entry count in every row ...
    assert np.array_equal(ccol, col_h[keep])                                     # ... at the same columns
    assert np.array_equal(cx, out[keep].astype(np.int64))                        # every count, bit-exact
    assert int(cx.sum()) == O.fast_tricount(rp_h, col_h) and cx.max() > 1000
    del C, L, ones, cx, out
    vals = rmat.values_torch(nnz, dev, seed=43)
    Lf = gb.Matrix.from_csr(gb.FP64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    F = Lf.mxm(Lf, semiring=gb.FP64.PLUS_TIMES, mask=Lf)
    frp, fcol, fx = F.to_csr()
    outf, hasf = O.fast_masked_mxm(rp_h, col_h, vals.cpu().numpy())
    assert np.array_equal(hasf, has) and np.array_equal(frp, crp) and np.array_equal(fcol, ccol)
    assert np.allclose(fx, outf[keep], rtol=1e-6, atol=0.0)


@pytest.mark.parametrize("typ,rtol,atol", [("FP32", 1e-4, 1e-3), ("FP64", 1e-9, 1e-9)])
def test_batched_bc_rmat22_against_the_oracle(gb, torch_dev, typ, rtol, atol):
    """The whole batched betweenness centrality of gap/bcmark.py:16-67 at R-MAT-22, ns = 4 (round 3 timed it, nothing checked it): depth
    and the entry count of every level's frontier equal to the oracle's (exact), every centrality value to 1e-4 — the driver computes
    in FP32, the oracle restates it in doubles; path counts pass 2^24 at this size.  Round 6: the same driver in FP64 against the same
    oracle to 1e-9 — the algorithm itself is inside the north star's 1e-6, the 1e-4 above is FP32's."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from bc_algorithm import bc
    torch, dev = torch_dev
    from pygraphblas_amd import rmat
    n = 1 << SCALE
    rowptr, col = rmat.csr_torch(SCALE, dev, seed=42, drop_self_loops=True)                       # directed
    trp, tcol = rmat.csr_torch(SCALE, dev, seed=42, drop_self_loops=True, transpose=True)
    nnz = int(col.numel())
    ones = torch.ones(nnz, dtype=torch.float32 if typ == "FP32" else torch.float64, device=dev)
    A = gb.Matrix.from_csr(TYPE[typ], n, n, rowptr.data_ptr(), col.data_ptr(), (ones.data_ptr(), nnz), device=True)
    AT = gb.Matrix.from_csr(TYPE[typ], n, n, trp.data_ptr(), tcol.data_ptr(), (ones.data_ptr(), nnz), device=True)
    deg = (rowptr[1:] - rowptr[:-1])
    sources = [int(x) for x in torch.argsort(deg, descending=True, stable=True)[:4].cpu()]
    sizes = []
    cent, depth = bc(gb, sources, AT, A, sizes=sizes, typ=TYPE[typ])
    got = cent.to_dense_arrays()[0].astype(np.float64)
    want, odepth, osizes = O.fast_bc(rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32), trp.cpu().numpy().view(np.uint32), tcol.cpu().numpy().view(np.uint32), sources)
    assert depth == odepth and depth >= 4
    assert sizes == osizes                                                       # the frontiers' patterns have the oracle's sizes, level by level
    assert want.max() > 1e3
    assert np.allclose(got, want, rtol=rtol, atol=atol), float(np.abs(got - want).max())


# ---- configs[4], single-GPU step ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mtype", ["FP32", "BOOL"])
def test_config4_rmat22_pagerank_step_fp32(gb, torch_dev, mtype):
    """r<accum PLUS> += A' (PLUS_SECOND) w — the product of gap/prmark.py:22-23.  With mtype BOOL the matrix is the pattern
    matrix the reference's driver builds (gap/prmark.py:47): its stored values are 1 byte wide while the semiring computes
    in FP32 — the case ADVICE.md (round 1) found reading past the value array in kernel X's plan builder."""
    torch, dev = torch_dev
    from pygraphblas_amd import rmat, descriptor as D
    n = 1 << SCALE
    rowptr, col = rmat.csr_torch(SCALE, dev, seed=42)
    nnz = int(col.numel())
    assert nnz >= 1 << 22
    if mtype == "FP32":
        vals = torch.ones(nnz, dtype=torch.float32, device=dev)
    else:
        vals = torch.ones(nnz, dtype=torch.bool, device=dev)
    A = gb.Matrix.from_csr(TYPE[mtype], n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    ws = rmat.values_torch(n, dev, seed=45, dtype=torch.float32)
    w = gb.Vector.from_dense_array((ws.data_ptr(), n), gb.FP32, device=True)
    teleport = np.float32(0.15 / n)
    r = gb.Vector.dense(gb.FP32, n, fill=float(teleport))
    A.mxv(w, out=r, accum=gb.FP32.PLUS, semiring=gb.FP32.PLUS_SECOND, desc=D.T0)
    gr, gp = r.to_dense_arrays()
    assert gp.all()
    # oracle: y = A' (PLUS_SECOND) w on the CSR of the transpose, then r = teleport + y where y has an entry
    import scipy.sparse as sp
    rp = rowptr.cpu().numpy().view(np.uint32).astype(np.int64); ci = col.cpu().numpy().view(np.uint32).astype(np.int64)
    At = sp.csr_matrix((np.ones(nnz, np.float32), ci, rp), shape=(n, n)).T.tocsr()
    # (row sums formed in double and rounded once: the reference's own summation order is unspecified, and a sequential
    #  float sum over a hub's 1.6e5 terms is itself ~1e-5 off — SURVEY.md §8c)
    y, pres = O.fast_spmv(At.indptr.astype(np.uint32), At.indices.astype(np.uint32), None, ws.cpu().numpy(), semiring="PLUS_SECOND_WIDE")
    exp = np.where(pres != 0, teleport + y, teleport).astype(np.float32)
    assert np.allclose(gr, exp, rtol=1e-6, atol=0.0)
    y64 = At.astype(np.float64) @ ws.cpu().numpy().astype(np.float64)          # second opinion: scipy in FP64
    assert np.allclose(gr, teleport + y64, rtol=1e-6, atol=0.0)


def test_config4_rmat22_pagerank_loop_with_dangling_vertices(gb, torch_dev):
    """The whole loop of gap/prmark.py:8-30 (world of one, pygraphblas_amd.dist.pagerank): w = t / d has no entry for the
    dangling vertices (half of R-MAT-22's), r is full and accumulates with PLUS — the product may then treat the holes of w as
    the monoid's identity and run the full-operand kernel.  Compared with the same iteration in FP64 (scipy): same number
    of iterations, every rank value within 1e-6."""
    torch, dev = torch_dev
    import scipy.sparse as sp
    from pygraphblas_amd import rmat, dist as gdist
    n = 1 << SCALE
    rowptr, col = rmat.csr_torch(SCALE, dev, seed=42, transpose=True)                    # rows of A'
    nnz = int(col.numel())
    ones = torch.ones(nnz, dtype=torch.float32, device=dev)
    At = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (ones.data_ptr(), nnz), device=True)
    none = gb.Matrix.sparse(gb.FP32, n, n)
    rp, ci = rowptr.cpu().numpy().view(np.uint32).astype(np.int64), col.cpu().numpy().view(np.uint32).astype(np.int64)
    deg = np.bincount(ci, minlength=n).astype(np.float64)                                # out-degree of j = entries in column j of A'
    assert (deg == 0).sum() > n // 4                                                     # plenty of dangling vertices
    d = gb.Vector.from_arrays(np.flatnonzero(deg).astype(np.uint64), deg[deg > 0].astype(np.float32), n, gb.FP32)
    r, its, rdiff = gdist.pagerank(gdist.Comm(0, 1), At, none, d, n, [0, n])
    gr, gp = r.to_dense_arrays()
    assert gp.all()
    M = sp.csr_matrix((np.ones(nnz), ci, rp), shape=(n, n))
    dd = np.where(deg > 0, deg / 0.85, np.nan); rr = np.full(n, 1.0 / n); tt = np.zeros(n); k = 0
    for i in range(100):
        tt, rr = rr, tt
        w = np.where(deg > 0, tt / dd, 0.0)
        rr = (1 - 0.85) / n + M @ w
        k = i + 1
        if np.abs(tt - rr).sum() <= 1e-4:
            break
    assert its == k
    assert np.allclose(gr, rr, rtol=1e-6, atol=0.0)


def test_accumulate_into_a_full_vector_in_the_merge_kernel(gb, torch_dev):
    """`w += A (+).(x) u` with the monoid's own operator into a resident full w, no mask: kernel X's merge applies the accumulator in
    its store (k_xp_merge<T, SR, 1>: rows with entries become w (+) sum in place, the others stay, no tval, no epilogue kernel) — and
    when w is a fill that was never written (`w[:] = s`, non-blocking mode) the fill folds into that store too (k_xp_merge<T, SR, 2>).
    R-MAT-22 FP64 PLUS_TIMES against the oracle's loop; both forms twice (the second call accumulates onto the first's result)."""
    torch, dev = torch_dev
    from pygraphblas_amd import rmat
    n = 1 << SCALE
    rowptr, col = rmat.csr_torch(SCALE, dev, seed=42)
    nnz = int(col.numel())
    vals = rmat.values_torch(nnz, dev, seed=43)
    xs = rmat.values_torch(n, dev, seed=44)
    w0 = rmat.values_torch(n, dev, seed=48)
    A = gb.Matrix.from_csr(gb.FP64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    x = gb.Vector.from_dense_array((xs.data_ptr(), n), gb.FP64, device=True)
    y, pres = O.fast_spmv(rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32), vals.cpu().numpy(), xs.cpu().numpy())
    y = np.where(pres != 0, y, 0.0)
    # (1) a resident full vector
    w = gb.Vector.from_dense_array((w0.data_ptr(), n), gb.FP64, device=True)
    A.mxv(x, semiring=gb.FP64.PLUS_TIMES)                          # (the matrix's first product runs kernel W; what follows is about kernel X's store modes)
    A.mxv(x, out=w, accum=gb.FP64.PLUS, semiring=gb.FP64.PLUS_TIMES)
    assert "k_spmv_xcd" in gb.last_kernel_plan()
    g1, p1 = w.to_dense_arrays()
    assert p1.all() and np.allclose(g1, w0.cpu().numpy() + y, rtol=1e-6, atol=0.0)
    A.mxv(x, out=w, accum=gb.FP64.PLUS, semiring=gb.FP64.PLUS_TIMES)
    g2, p2 = w.to_dense_arrays()
    assert p2.all() and np.allclose(g2, w0.cpu().numpy() + 2 * y, rtol=1e-6, atol=0.0)
    # (2) a pending fill
    w = gb.Vector.sparse(gb.FP64, n)
    w[:] = 0.25
    A.mxv(x, out=w, accum=gb.FP64.PLUS, semiring=gb.FP64.PLUS_TIMES)
    g3, p3 = w.to_dense_arrays()
    assert p3.all() and np.allclose(g3, 0.25 + y, rtol=1e-6, atol=0.0)
    assert w.nvals == n
    # (3) another accumulator than the monoid's: the general epilogue
    w = gb.Vector.from_dense_array((w0.data_ptr(), n), gb.FP64, device=True)
    A.mxv(x, out=w, accum=gb.FP64.MAX, semiring=gb.FP64.PLUS_TIMES)
    g4, _ = w.to_dense_arrays()
    assert np.allclose(g4, np.where(pres != 0, np.maximum(w0.cpu().numpy(), y), w0.cpu().numpy()), rtol=1e-6, atol=0.0)


# ---- configs[4] at its STATED size: R-MAT scale-25 on one MI355X ------------------------------------------------------------
def test_config4_rmat25_pagerank_single_gpu(gb, torch_dev):
    """FP32 PageRank of gap/prmark.py:8-30 on R-MAT scale-25 (n = 33 554 432, 16·2^25 sampled edges) held by ONE MI355X — the
    single-GPU anchor of configs[4]'s 1→8 curve.  (a) one product r<accum PLUS> += A' (PLUS_SECOND) w with desc T0 against the
    oracle's loop with double row sums (1e-6); (b) the whole loop against the same iteration carried in FP64 on the host
    (oracle fast_spmv PLUS_SECOND on doubles): same number of iterations, every rank value within 1e-6."""
    torch, dev = torch_dev
    from pygraphblas_amd import rmat, loops, descriptor as D
    S = 25
    n = 1 << S
    rowptr, col = rmat.csr_torch(S, dev, seed=42)                                        # A
    nnz = int(col.numel())
    assert nnz > 5 * 10**8
    ones = torch.ones(nnz, dtype=torch.float32, device=dev)
    A = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (ones.data_ptr(), nnz), device=True)
    deg_t = (rowptr[1:] - rowptr[:-1])
    deg = deg_t.cpu().numpy().astype(np.float64)                                         # out-degrees = row lengths of A
    del rowptr, col, ones, deg_t
    torch.cuda.empty_cache()
    trp, tcol = rmat.csr_torch(S, dev, seed=42, transpose=True)                          # rows of A' for the host-side oracle
    assert int(tcol.numel()) == nnz
    rp = trp.cpu().numpy().view(np.uint32); ci = tcol.cpu().numpy().view(np.uint32)
    del trp, tcol
    torch.cuda.empty_cache()
    # (a) one step
    ws = rmat.values_torch(n, dev, seed=45, dtype=torch.float32)
    w = gb.Vector.from_dense_array((ws.data_ptr(), n), gb.FP32, device=True)
    teleport = np.float32(0.15 / n)
    r = gb.Vector.dense(gb.FP32, n, fill=float(teleport))
    A.mxv(w, out=r, accum=gb.FP32.PLUS, semiring=gb.FP32.PLUS_SECOND, desc=D.T0)
    plan = gb.last_kernel_plan()
    assert "k_spmv_xcd" in plan or "k_spmv_wavepipe" in plan, plan                      # a full-operand pipeline kernel, not the row-block fallback
    gr, gp = r.to_dense_arrays()
    assert gp.all()
    y, pres = O.fast_spmv(rp, ci, None, ws.cpu().numpy(), semiring="PLUS_SECOND_WIDE")
    exp = np.where(pres != 0, teleport + y, teleport).astype(np.float32)
    assert np.allclose(gr, exp, rtol=1e-6, atol=0.0)
    del w, r, ws, gr, gp, y, exp
    # (b) the loop
    assert (deg == 0).sum() > n // 4
    d = gb.Vector.from_dense_array(deg.astype(np.float32), gb.FP32, present=(deg > 0).astype(np.uint8))
    r, its, rdiff = loops.pagerank(A, d)
    gr, gp = r.to_dense_arrays()
    assert gp.all()
    dd = np.where(deg > 0, deg / 0.85, 1.0); rr = np.full(n, 1.0 / n); tt = np.zeros(n); k = 0
    for i in range(100):
        tt, rr = rr, tt
        wv = np.where(deg > 0, tt / dd, 0.0)
        yy, _ = O.fast_spmv(rp, ci, None, wv, semiring="PLUS_SECOND")
        rr = (1 - 0.85) / n + yy
        k = i + 1
        if np.abs(tt - rr).sum() <= 1e-4:
            break
    assert its == k
    assert np.allclose(gr, rr, rtol=1e-6, atol=0.0)


# ---- MIN_PLUS at scale: the reference's shortest-path loop on R-MAT-22 ------------------------------------------------------
@pytest.mark.parametrize("wtype", ["INT64", "FP64"])
def test_min_plus_sssp_rmat22(gb, torch_dev, wtype):
    """`v<accum MIN> = v MIN_PLUS A` repeated until nothing changes (demo/Intro-Prez.ipynb:1034-1045, pygraphblas/vector.py:883-885)
    on the directed R-MAT-22 with INT64 weights in [1, 255] / FP64 weights in (0, 1]: distances bit-exact (INT64) / 1e-6 (FP64)
    against the oracle's loop — sweep count included — and against scipy.sparse.csgraph.dijkstra; the first sweeps must run the
    push kernel (a one-entry operand), the late ones a pull kernel (the operand holds a third of the vertices)."""
    torch, dev = torch_dev
    import scipy.sparse as sp
    from scipy.sparse.csgraph import dijkstra
    from pygraphblas_amd import rmat, loops
    n = 1 << SCALE
    rowptr, col = rmat.csr_torch(SCALE, dev, seed=42, drop_self_loops=True)
    nnz = int(col.numel())
    u = rmat.values_torch(nnz, dev, seed=47)
    if wtype == "INT64":
        vals = (u * 255.0).to(torch.int64) + 1
    else:
        vals = 1.0 - u                                                                   # (0, 1]
    A = gb.Matrix.from_csr(TYPE[wtype], n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    src = int(torch.argmax(rowptr[1:] - rowptr[:-1]))
    plans = []
    v, sweeps = loops.sssp(A, src, plans=plans)
    gd, gp = v.to_dense_arrays()
    rp = rowptr.cpu().numpy().view(np.uint32); ci = col.cpu().numpy().view(np.uint32); hv = vals.cpu().numpy()
    dist, pres, osweeps = O.fast_sssp(rp, ci, hv, src)
    assert sweeps == osweeps
    assert np.array_equal(gp != 0, pres != 0)
    assert int(pres.sum()) > n // 4
    if wtype == "INT64":
        assert np.array_equal(gd[pres != 0], dist[pres != 0])                            # bit-exact
    else:
        assert np.allclose(gd[pres != 0], dist[pres != 0], rtol=1e-6, atol=0.0)
    assert plans[0] == "k_spmspv_push", plans
    assert plans[-1].startswith("k_spmv_"), plans
    G = sp.csr_matrix((hv.astype(np.float64), ci.astype(np.int32), rp.astype(np.int64)), shape=(n, n))
    dj = dijkstra(G, directed=True, indices=src)
    assert np.array_equal(np.isfinite(dj), pres != 0)
    if wtype == "INT64":
        assert np.array_equal(dj[pres != 0].astype(np.int64), gd[pres != 0])
    else:
        assert np.allclose(gd[pres != 0], dj[pres != 0], rtol=1e-6, atol=0.0)


def test_holes_filled_with_the_identity_only_when_the_pattern_cannot_matter(gb, gpu):
    """Small cases around the fast path above: with a sparse output, a mask, or another accumulator the product must keep
    the exact pattern semantics (oracle), and with a full output + same-operator accumulator the values must agree."""
    rng = np.random.default_rng(11)
    n = 300
    flat = np.sort(rng.choice(n * n, 4000, replace=False)).astype(np.uint64)
    I, J = np.divmod(flat, np.uint64(n)); X = rng.integers(1, 5, len(flat)).astype(np.float32)
    A = gb.Matrix.from_arrays(I, J, X, n, n, gb.FP32)
    ui = np.sort(rng.choice(n, 120, replace=False)).astype(np.uint64); ux = rng.integers(1, 9, 120).astype(np.float32)
    u = gb.Vector.from_arrays(ui, ux, n, gb.FP32)
    At = O.Tuples("FP32", n, n, I, J, X); uo = O.col_vector("FP32", n, ui, ux)
    for full_out in (True, False):
        for acc in ("PLUS", "MIN"):
            wi = np.arange(n, dtype=np.uint64) if full_out else np.sort(rng.choice(n, 50, replace=False)).astype(np.uint64)
            wx = rng.integers(1, 9, len(wi)).astype(np.float32)
            if full_out:
                w = gb.Vector.dense(gb.FP32, n, fill=0.0); w += gb.Vector.from_arrays(wi, wx, n, gb.FP32)      # full, resident in HBM
            else:
                w = gb.Vector.from_arrays(wi, wx, n, gb.FP32)
            A.mxv(u, out=w, accum=getattr(gb.FP32, acc), semiring=gb.FP32.PLUS_SECOND)
            exp = O.mxv(O.col_vector("FP32", n, wi, wx), At, uo, "PLUS", "SECOND", "FP32", accum=acc)
            gi, gx = w.to_arrays()
            assert np.array_equal(gi, exp.I), (full_out, acc)
            assert np.allclose(gx, exp.X, rtol=1e-6, atol=0.0), (full_out, acc)


def test_config1_per_panel_merge_fallback_in_a_fresh_process(gpu):
    """Kernel X keeps its per-panel merge kernel (k_xp_combine) for matrices whose rows have more sub-rows than a block of the
    row-major merge holds; GRB_MI355X_XP_OLD_MERGE=1 selects it (read once per process, hence the subprocess).  Same check as
    configs[1]: R-MAT-22 FP64 PLUS_TIMES against the oracle's loop, and bit-identical results call after call."""
    import os, subprocess, sys
    code = r"""
import numpy as np, torch, sys
sys.path.insert(0, %r)
import pygraphblas_amd as gb
from pygraphblas_amd import rmat
from oracle import oracle as O
S = 22; n = 1 << S; dev = torch.device("cuda", 0)
rowptr, col = rmat.csr_torch(S, dev, seed=42); nnz = int(col.numel())
vals = rmat.values_torch(nnz, dev, seed=43); xs = rmat.values_torch(n, dev, seed=44)
A = gb.Matrix.from_csr(gb.FP64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
x = gb.Vector.from_dense_array((xs.data_ptr(), n), gb.FP64, device=True)
A.mxv(x, semiring=gb.FP64.PLUS_TIMES)
w = A.mxv(x, semiring=gb.FP64.PLUS_TIMES); assert "k_spmv_xcd" in gb.last_kernel_plan()
gy, gp = w.to_dense_arrays()
y, pres = O.fast_spmv(rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32), vals.cpu().numpy(), xs.cpu().numpy())
assert np.array_equal(gp != 0, pres != 0) and np.allclose(gy[pres != 0], y[pres != 0], rtol=1e-6, atol=0.0)
g2, _ = A.mxv(x, semiring=gb.FP64.PLUS_TIMES).to_dense_arrays(); assert np.array_equal(g2[pres != 0], gy[pres != 0])
print("OK old merge")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, GRB_MI355X_XP_OLD_MERGE="1"))
    assert r.returncode == 0 and "OK old merge" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
