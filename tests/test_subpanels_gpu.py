"""Kernel X's plans with column sub-panels (round 4; pygraphblas_amd/csrc/grb_spmv_xcd.hpp: XcdPlan::S / own, k_xp_merge_wide) at a size the
suite runs in seconds — the library takes them by itself only from 2^23 columns on (tests/test_baseline_configs_gpu.py covers that at
R-MAT-25).  `GRB_MI355X_XS` forces S sub-panels per XCD at plan time, `GRB_MI355X_XOWN` chooses a table per sub-panel (1) or per XCD (0).

For every (S, table mode, type): the product against the oracle (integer-valued data: bit-exact in every type, whatever order the
sub-rows add in) and every store mode of the merge kernel — plain, accumulate into a resident full vector (EPI 1), a pending fill folded
into the store (EPI 2), threshold + accumulate MIN over an operand with holes (EPI 3: the sweeps of the reference's shortest-path loop,
demo/Intro-Prez.ipynb:1034-1045).  Reference call sites: pygraphblas/matrix.py:2714-2725 (mxv), pygraphblas/vector.py:960-970 (vxm)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
SCALE = 19


@pytest.fixture(scope="module")
def graph(gpu):
    from pygraphblas_amd import rmat
    rp, col = rmat.csr_numpy(SCALE)
    assert len(col) >= 1 << 22                                  # large enough for kernel X by the dispatcher's own rule
    return rp, col


def _plan(gb):
    return gb.last_kernel_plan()


@pytest.mark.parametrize("S,own", [(2, 1), (4, 1), (8, 1), (2, 0), (4, 0)])
@pytest.mark.parametrize("typ", ["FP64", "FP32", "INT64"])
def test_products_on_sub_panel_plans_match_the_oracle(gb, graph, monkeypatch, S, own, typ):
    rp, col = graph
    n = 1 << SCALE
    monkeypatch.setenv("GRB_MI355X_XS", str(S)); monkeypatch.setenv("GRB_MI355X_XOWN", str(own))
    T = getattr(gb, typ); npt = T._np
    rng = np.random.default_rng(S * 10 + own)
    vals = rng.integers(1, 8, len(col)).astype(npt)
    xs = rng.integers(0, 4, n).astype(npt)
    A = gb.Matrix.from_csr(T, n, n, rp, col, vals)
    x = gb.Vector.from_dense_array(xs, T)
    y, pres = O.fast_spmv(rp, col, vals.astype(np.float64), xs.astype(np.float64))           # exact: small integers
    tag = f"subpanels={S}" + ("/own-tables" if own else "")
    A.mxv(x, semiring=T.PLUS_TIMES)                                                         # (a matrix's first product may run kernel W: the plan policy)
    w = A.mxv(x, semiring=T.PLUS_TIMES)
    assert "k_spmv_xcd" in _plan(gb) and tag in _plan(gb), _plan(gb)
    gy, gp = w.to_dense_arrays()
    assert np.array_equal(gp != 0, pres != 0) and np.array_equal(gy[pres != 0].astype(np.float64), y[pres != 0])
    # EPI 1: accumulate into a resident full vector with the monoid's operator
    w0 = rng.integers(0, 5, n).astype(npt)
    r = gb.Vector.from_dense_array(w0, T)
    A.mxv(x, out=r, accum=T.PLUS, semiring=T.PLUS_TIMES)
    assert tag in _plan(gb)
    g1, p1 = r.to_dense_arrays()
    assert p1.all() and np.array_equal(g1.astype(np.float64), w0.astype(np.float64) + np.where(pres != 0, y, 0.0))
    # EPI 2: `r[:] = c` pending, folded into the product's store (gap/prmark.py:21-23)
    r2 = gb.Vector.sparse(T, n)
    r2[:] = 3
    A.mxv(x, out=r2, accum=T.PLUS, semiring=T.PLUS_TIMES)
    g2, p2 = r2.to_dense_arrays()
    assert p2.all() and np.array_equal(g2.astype(np.float64), 3.0 + np.where(pres != 0, y, 0.0))
    # pattern-only multiply (PLUS_SECOND: the plan stores no values)
    B = gb.Matrix.from_csr(T, n, n, rp, col, vals)
    B.mxv(x, semiring=T.PLUS_SECOND)
    ws = B.mxv(x, semiring=T.PLUS_SECOND)
    assert tag in _plan(gb)
    ys, ps = O.fast_spmv(rp, col, np.ones(len(col)), xs.astype(np.float64))
    gs, gps = ws.to_dense_arrays()
    assert np.array_equal(gps != 0, ps != 0) and np.array_equal(gs[ps != 0].astype(np.float64), ys[ps != 0])


@pytest.mark.parametrize("S,own", [(2, 1), (4, 1), (4, 0)])
@pytest.mark.parametrize("typ", ["INT64", "FP64"])
def test_min_plus_sweeps_on_sub_panel_plans(gb, graph, monkeypatch, S, own, typ):
    """`v<accum MIN> = v MIN_PLUS A` over an operand with holes: the "big holes" product with the threshold and the accumulator in the
    wide merge's store (EPI 3) — three sweeps from a third of the vertices, each against the oracle's generic product."""
    rp, col = graph
    n = 1 << SCALE
    monkeypatch.setenv("GRB_MI355X_XS", str(S)); monkeypatch.setenv("GRB_MI355X_XOWN", str(own))
    T = getattr(gb, typ); npt = T._np
    rng = np.random.default_rng(7 + S)
    wts = rng.integers(1, 256, len(col)).astype(npt)
    A = gb.Matrix.from_csr(T, n, n, rp, col, wts)
    idx = np.sort(rng.choice(n, n // 3, replace=False)).astype(np.uint64)
    d0 = rng.integers(0, 1000, len(idx)).astype(npt)
    v = gb.Vector.from_arrays(idx, d0, n, T)
    rows = np.repeat(np.arange(n, dtype=np.uint64), np.diff(rp.astype(np.int64)))
    At = O.Tuples(typ, n, n, rows, col, wts)
    ov = O.row_vector(typ, n, idx, d0)
    seen = False
    for sweep in range(3):
        v.vxm(A, semiring=T.MIN_PLUS, accum=T.MIN, out=v)
        seen = seen or (f"subpanels={S}" in _plan(gb))
        ov = O.vxm(ov, ov, At, "MIN", "PLUS", typ, accum="MIN")
        gi, gx = v.to_arrays()
        assert np.array_equal(gi, ov.J) and np.array_equal(gx, ov.X), (sweep, _plan(gb))
    assert seen                                                                              # kernel X on a sub-panel plan ran at least one of the sweeps


@pytest.mark.parametrize("S", [1, 4])
@pytest.mark.parametrize("typ,sr", [("FP64", "PLUS_TIMES"), ("FP32", "PLUS_SECOND"), ("INT64", "MIN_PLUS"), ("UINT32", "PLUS_TIMES")])
def test_lane_per_piece_layout_matches_the_tile_pipeline(gb, graph, monkeypatch, S, typ, sr):
    """Round 6's layout experiment (grb_spmv_sell.hpp, `GRB_MI355X_SELL=1`: a lane per piece of a sub-row, chunks of 64 pieces sorted by length inside
    windows, the steps' records in two spare bits of the column words) against the oracle — integer-valued data: bit-exact in every type whatever order
    the pieces add in — with one stream per XCD and with four sub-panels per XCD (a table per sub-panel: the kernel walks an XCD's streams one after the
    other), on the plain store and on the accumulate-into-a-fill store of gap/prmark.py:21-23, twice for the same bits.  The experiment's verdict
    (slower than the tile pipeline: profiles/r06_spmv_layout_experiment.txt) does not depend on this test; its correctness does."""
    rp, col = graph
    n = 1 << SCALE
    monkeypatch.setenv("GRB_MI355X_SELL", "1"); monkeypatch.setenv("GRB_MI355X_XS", str(S)); monkeypatch.setenv("GRB_MI355X_XOWN", "1")
    T = getattr(gb, typ); npt = T._np
    rng = np.random.default_rng(77 + S)
    vals = rng.integers(1, 8, len(col)).astype(npt)
    xs = rng.integers(1, 5, n).astype(npt)
    A = gb.Matrix.from_csr(T, n, n, rp, col, vals)
    x = gb.Vector.from_dense_array(xs, T)
    semiring = getattr(T, sr)
    if sr == "MIN_PLUS":
        rows = np.repeat(np.arange(n), np.diff(rp.astype(np.int64)))
        y = np.full(n, np.iinfo(np.int64).max, np.int64); np.minimum.at(y, rows, vals.astype(np.int64) + xs[col.astype(np.int64)].astype(np.int64))
        pres = (np.diff(rp.astype(np.int64)) > 0).astype(np.uint8)
    else:
        y, pres = O.fast_spmv(rp, col, vals.astype(np.float64) if sr == "PLUS_TIMES" else None, xs.astype(np.float64), **({} if sr == "PLUS_TIMES" else {"semiring": "PLUS_SECOND"}))
    A.mxv(x, semiring=semiring)
    w = A.mxv(x, semiring=semiring)
    assert "lane-per-piece" in _plan(gb), _plan(gb)
    gy, gp = w.to_dense_arrays()
    assert np.array_equal(gp != 0, pres != 0) and np.array_equal(gy[pres != 0].astype(np.float64), np.asarray(y)[pres != 0].astype(np.float64))
    again = A.mxv(x, semiring=semiring).to_dense_arrays()[0]
    assert np.array_equal(again.view(np.uint8), gy.view(np.uint8))
    if sr != "MIN_PLUS":
        r2 = gb.Vector.sparse(T, n)
        r2[:] = 3
        A.mxv(x, out=r2, accum=T.PLUS, semiring=semiring)
        g2, p2 = r2.to_dense_arrays()
        assert p2.all() and np.array_equal(g2.astype(np.float64), 3.0 + np.where(pres != 0, y, 0.0))
    # the same matrix object with the layout switched off at run time: the tile pipeline on the plan that carries both
    monkeypatch.setenv("GRB_MI355X_SELL_RUN", "0")
    t = A.mxv(x, semiring=semiring).to_dense_arrays()[0]
    assert np.array_equal(t[pres != 0].astype(np.float64), np.asarray(y)[pres != 0].astype(np.float64))
