"""GPU parity of GrB_mxv / GrB_vxm (HIP kernels, through the C ABI) against the CPU oracle.

Bit-exact for BOOL / integer types; floating point is compared at rtol 1e-6 (the tolerance
BASELINE.json's north_star states) — and exactly where the test data sits on a 1/8 grid.
Cases cover what the reference's tests cover for this path (tests/test_vector.py:298-315,
tests/test_matrix.py:293-306, tests/test_descriptor.py:13-30): masks (valued / structural /
complemented / empty), replace, accum, transposed inputs, output aliasing an input, typecast of
operands and output, empty and ragged inputs, rows longer than one kernel block.
"""
import itertools
import os

import numpy as np
import pytest

from oracle import oracle as O
import pygraphblas_amd as gb
from pygraphblas_amd import descriptor as D
from helpers import TYPE, rand_matrix, rand_vector, to_matrix, to_vector, vector_pairs, assert_same

pytestmark = pytest.mark.gpu

SEMIRINGS = {
    "BOOL": ["LOR_LAND", "ANY_PAIR", "LXOR_LAND", "LAND_LOR", "EQ_LOR"],
    "INT": ["PLUS_TIMES", "MIN_PLUS", "PLUS_PAIR", "PLUS_SECOND", "PLUS_FIRST", "MAX_MIN", "MIN_FIRST", "PLUS_PLUS", "TIMES_PLUS",
            "PLUS_LAND", "MAX_MINUS", "MIN_RDIV", "PLUS_ISGT", "ANY_PAIR"],
    "FP": ["PLUS_TIMES", "MIN_PLUS", "PLUS_PAIR", "PLUS_SECOND", "PLUS_FIRST", "MAX_TIMES", "MIN_MAX", "PLUS_MINUS", "MIN_DIV"],
}


def family(t):
    return "BOOL" if t == "BOOL" else ("FP" if t.startswith("FP") else "INT")


def run_case(rng, typ, sr_name, nrows, ncols, dens, udens, *, vxm=False, tran=False, mask=None, accum=None, replace=False,
             out_typ=None, a_typ=None, u_typ=None, method=None):
    add, mul = sr_name.split("_")
    a_typ, u_typ, out_typ = a_typ or typ, u_typ or typ, out_typ or typ
    # stored matrix A is (nrows x ncols); the effective operand may be transposed
    A = rand_matrix(rng, a_typ, nrows, ncols, dens)
    eff_r, eff_c = (ncols, nrows) if tran else (nrows, ncols)
    n_in, n_out = (eff_r, eff_c) if vxm else (eff_c, eff_r)
    ui, ux = rand_vector(rng, u_typ, n_in, udens)
    wi, wx = rand_vector(rng, out_typ, n_out, 0.4)
    mi = mx = None
    mtyp = None
    if mask is not None:
        mtyp = mask["typ"]
        mi, mx = rand_vector(rng, mtyp, n_out, mask.get("dens", 0.5))
    sr = getattr(TYPE[typ], sr_name)
    gA, gu, gw = to_matrix(A), to_vector(u_typ, n_in, ui, ux), to_vector(out_typ, n_out, wi, wx)
    gm = to_vector(mtyp, n_out, mi, mx) if mask is not None else None
    flags = []
    if replace:
        flags.append("R")
    if mask is not None and mask.get("struct"):
        flags.append("S")
    if mask is not None and mask.get("comp"):
        flags.append("C")
    if tran:
        flags.append("T1" if vxm else "T0")
    desc = getattr(D, "".join(flags)) if flags else None
    acc = getattr(TYPE[out_typ], accum) if accum else None
    if method:
        os.environ["GRB_MI355X_SPMV"] = method
    try:
        if vxm:
            gu.vxm(gA, semiring=sr, out=gw, mask=gm, accum=acc, desc=desc)
        else:
            gA.mxv(gu, semiring=sr, out=gw, mask=gm, accum=acc, desc=desc)
    finally:
        os.environ.pop("GRB_MI355X_SPMV", None)
    kw = dict(accum=accum, accum_type=out_typ, replace=replace, mask_comp=bool(mask and mask.get("comp")),
              mask_struct=bool(mask and mask.get("struct")))
    if vxm:
        exp = O.vxm(O.row_vector(out_typ, n_out, wi, wx), O.row_vector(u_typ, n_in, ui, ux), A, add, mul, typ,
                    mask=O.row_vector(mtyp, n_out, mi, mx) if mask is not None else None, tran_a=tran, **kw)
        exp_idx = exp.J
    else:
        exp = O.mxv(O.col_vector(out_typ, n_out, wi, wx), A, O.col_vector(u_typ, n_in, ui, ux), add, mul, typ,
                    mask=O.col_vector(mtyp, n_out, mi, mx) if mask is not None else None, tran_a=tran, **kw)
        exp_idx = exp.I
    gi, gx = vector_pairs(gw)
    # quotients are not on the 1/8 grid: the order of summation matters in the last bit -> 1e-6 relative
    rtol = 1e-6 if ("DIV" in sr_name and out_typ.startswith("FP")) else 0.0
    assert_same(out_typ, gi, gx, exp_idx, exp.X, rtol=rtol, what=f"{typ}.{sr_name} vxm={vxm} tran={tran} mask={mask} accum={accum} plan={gb.last_kernel_plan()}")


ALL = ["BOOL", "INT8", "UINT8", "INT16", "UINT16", "INT32", "UINT32", "INT64", "UINT64", "FP32", "FP64"]


@pytest.mark.parametrize("typ", ALL)
def test_every_type_default_semirings(gpu, typ):
    rng = np.random.default_rng(hash(typ) % 2**32)
    for sr in SEMIRINGS[family(typ)]:
        for vxm in (False, True):
            run_case(rng, typ, sr, 37, 53, 0.15, 0.6, vxm=vxm)
            run_case(rng, typ, sr, 41, 29, 0.2, 1.0, vxm=vxm, tran=True)


@pytest.mark.parametrize("typ", ["BOOL", "INT64", "FP64", "UINT8", "FP32"])
@pytest.mark.parametrize("method", ["adaptive", "rowgroup", "push"])
def test_masks_accum_replace_all_kernels(gpu, typ, method):
    rng = np.random.default_rng(7)
    sr = {"BOOL": "LOR_LAND", "FP64": "PLUS_TIMES", "FP32": "PLUS_TIMES"}.get(typ, "MIN_PLUS" if method == "push" else "PLUS_TIMES")
    acc = {"BOOL": "LOR"}.get(typ, "PLUS")
    masks = [None, {"typ": "BOOL"}, {"typ": "BOOL", "comp": True}, {"typ": "INT32", "struct": True}, {"typ": "FP64", "comp": True, "struct": True},
             {"typ": "UINT8", "dens": 0.0, "comp": True}, {"typ": "BOOL", "dens": 1.0}]
    for mask, accum, replace, vxm in itertools.product(masks, [None, acc], [False, True], [False, True]):
        run_case(rng, typ, sr, 64, 48, 0.12, 0.3, vxm=vxm, mask=mask, accum=accum, replace=replace, method=method)


def test_typecasts(gpu):
    rng = np.random.default_rng(3)
    # operands of other types are cast into the semiring's domain; the result into the output's type
    run_case(rng, "BOOL", "LOR_LAND", 30, 30, 0.2, 0.5, a_typ="INT64", u_typ="UINT8", out_typ="BOOL", vxm=True)   # the BFS step
    run_case(rng, "FP64", "PLUS_TIMES", 30, 40, 0.2, 0.7, a_typ="FP32", u_typ="INT8", out_typ="FP32")
    run_case(rng, "INT64", "PLUS_TIMES", 30, 40, 0.2, 0.7, a_typ="UINT8", u_typ="INT16", out_typ="FP64", accum="PLUS")
    run_case(rng, "INT32", "MIN_PLUS", 25, 25, 0.3, 0.5, out_typ="INT8", mask={"typ": "FP32"}, accum="MIN")
    run_case(rng, "FP32", "PLUS_SECOND", 50, 50, 0.1, 1.0, a_typ="BOOL", accum="PLUS", tran=True)                  # the PageRank step


def test_edge_shapes(gpu):
    rng = np.random.default_rng(11)
    run_case(rng, "INT64", "PLUS_TIMES", 1, 1, 1.0, 1.0)
    run_case(rng, "INT64", "PLUS_TIMES", 5, 7, 0.0, 1.0)            # empty matrix
    run_case(rng, "INT64", "PLUS_TIMES", 5, 7, 0.5, 0.0)            # empty vector
    run_case(rng, "FP64", "PLUS_TIMES", 3, 9000, 0.9, 1.0)          # rows longer than one block (2048) and one part (8192)
    run_case(rng, "FP64", "PLUS_TIMES", 2, 30000, 0.95, 0.9)        # multi-part long rows with a bitmap operand
    run_case(rng, "INT32", "MIN_PLUS", 3000, 3, 0.5, 1.0)           # many short rows
    run_case(rng, "UINT16", "PLUS_TIMES", 2500, 2500, 0.002, 1.0)   # mostly empty rows
    run_case(rng, "BOOL", "LOR_LAND", 4, 20000, 0.9, 0.5, mask={"typ": "BOOL", "comp": True})


def test_output_aliases_input(gpu):
    # reference: tests/test_descriptor.py:13-30 (test_RCT0 / test_RC): out=w aliases the operand, empty mask complemented, replace
    M = gb.Matrix.from_lists([0, 1, 2], [1, 2, 0], [True, True, True])
    w = gb.Vector.sparse(gb.BOOL, 3); v = gb.Vector.sparse(gb.BOOL, 3)
    w[0] = True
    M.mxv(w, out=w, mask=v, desc=D.RCT0)
    assert w.iseq(gb.Vector.from_lists([1], [True], 3))
    w = gb.Vector.sparse(gb.BOOL, 3); w[0] = True
    M.mxv(w, out=w, mask=v, desc=D.RC)
    assert w.iseq(gb.Vector.from_lists([2], [True], 3))


def test_dimension_mismatch_raises(gpu):
    A = gb.Matrix.from_lists([0, 1], [1, 2], [1, 2], nrows=3, ncols=4)
    with pytest.raises(gb.DimensionMismatch):
        A.mxv(gb.Vector.from_lists([0], [1], size=3))
    with pytest.raises(gb.DimensionMismatch):
        gb.Vector.from_lists([0], [1], size=4).vxm(A)


def test_fp64_random_values_within_1e6(gpu):
    # unstructured double values: order of summation differs from the oracle's, tolerance 1e-6 relative
    rng = np.random.default_rng(5)
    A = rand_matrix(rng, "FP64", 2000, 2000, 0.01, small=False)
    ui, ux = rand_vector(rng, "FP64", 2000, 1.0, small=False)
    w = to_matrix(A).mxv(to_vector("FP64", 2000, ui, ux), semiring=gb.FP64.PLUS_TIMES)
    exp = O.mxv(O.col_vector("FP64", 2000), A, O.col_vector("FP64", 2000, ui, ux), "PLUS", "TIMES", "FP64")
    gi, gx = vector_pairs(w)
    assert_same("FP64", gi, gx, exp.I, exp.X, rtol=1e-6)


@pytest.mark.parametrize("method", ["wavepipe", "xcd"])
@pytest.mark.parametrize("typ,sr", [("FP64", "PLUS_TIMES"), ("FP32", "PLUS_SECOND"), ("INT64", "MIN_PLUS"), ("INT32", "PLUS_PAIR"), ("UINT32", "MAX_MIN")])
def test_wavepipe_kernel_shapes(gpu, typ, sr, method):
    """Kernels W / X (persistent merge-path pipeline + LDS hot table; X = one column panel per XCD) forced on shapes that stress the carries: rows that span
    tasks and whole waves' ranges, runs of empty rows, a single row, fewer tasks than waves, hot and cold columns."""
    rng = np.random.default_rng(17)
    shapes = [(3, 40000, 0.9),        # three rows of ~36000 entries: each spans ~70 tasks and several waves
              (1, 70000, 1.0),        # one row
              (6000, 300, 0.4),       # many medium rows
              (50000, 64, 0.02),      # mostly empty / tiny rows: row-marker dominated tasks
              (40, 30000, 0.3),       # rows of ~9000
              (2000, 2000, 0.001),    # very sparse: barely more than one task
              (20000, 16, 0.9),       # every column in one 128-byte line of u: kernel X puts all entries in one panel, seven panels are empty
              (3000, 40, 1.0),        # three lines of u: five empty panels, every entry served by the LDS tables
              (257, 4100, 0.25)]      # panels of ~32 000 entries: last tiles are partial
    for nrows, ncols, dens in shapes:
        run_case(rng, typ, sr, nrows, ncols, dens, 1.0, method=method)
        assert ("wavepipe" in gb.last_kernel_plan()) or ("xcd" in gb.last_kernel_plan()), gb.last_kernel_plan()
        if method == "xcd" and nrows * ncols * dens > 40000:
            assert "xcd" in gb.last_kernel_plan(), gb.last_kernel_plan()      # kernel X: one column panel per XCD
    # skewed columns (a few very hot ones) + a transposed operand through the cached CSC
    A = rand_matrix(rng, typ, 3000, 5000, 0.01)
    A.J[: len(A.J) // 2] = rng.integers(0, 7, len(A.J) // 2).astype(np.uint64)
    key = np.unique(A.I * np.uint64(5000) + A.J, return_index=True)[1]
    A = O.Tuples(typ, 3000, 5000, A.I[key], A.J[key], A.X[key])
    ui, ux = rand_vector(rng, typ, 5000, 1.0)
    os.environ["GRB_MI355X_SPMV"] = method
    try:
        w = to_matrix(A).mxv(to_vector(typ, 5000, ui, ux), semiring=getattr(TYPE[typ], sr))
    finally:
        os.environ.pop("GRB_MI355X_SPMV", None)
    add, mul = sr.split("_")
    exp = O.mxv(O.col_vector(typ, 3000), A, O.col_vector(typ, 5000, ui, ux), add, mul, typ)
    gi, gx = vector_pairs(w)
    assert_same(typ, gi, gx, exp.I, exp.X, what="wavepipe skewed")


@pytest.mark.parametrize("method", ["wavepipe", "xcd"])
def test_pipeline_results_are_bitwise_reproducible(gpu, method):
    """Chunks of the pipeline are handed to waves dynamically, but every chunk's arithmetic and the order in which partial sums are
    combined (carries, fix-up records, panel merge) are fixed: repeated FP64 products must agree bit for bit."""
    rng = np.random.default_rng(23)
    A = rand_matrix(rng, "FP64", 3000, 2500, 0.05, small=False)
    ui, ux = rand_vector(rng, "FP64", 2500, 1.0, small=False)
    M = to_matrix(A); u = to_vector("FP64", 2500, ui, ux)
    os.environ["GRB_MI355X_SPMV"] = method
    try:
        ref = None
        for _ in range(6):
            w = M.mxv(u, semiring=gb.FP64.PLUS_TIMES)
            gi, gx = vector_pairs(w)
            cur = (np.asarray(gi).tobytes(), np.asarray(gx, np.float64).tobytes())
            if ref is None: ref = cur
            assert cur == ref
        assert ("wavepipe" in gb.last_kernel_plan()) or ("xcd" in gb.last_kernel_plan()), gb.last_kernel_plan()
    finally:
        os.environ.pop("GRB_MI355X_SPMV", None)


def test_user_semirings_bitwise_run_and_math_multipliers_are_refused(gpu):
    """A semiring composed with GrB_Semiring_new from operators beyond FIRST..LXOR: the integer bitwise ones run in the
    kernels (checked against numpy), the math-library ones (POW, HYPOT, ...) are refused with a message instead of
    silently computing zeros (grb_opcommon.hpp)."""
    import ctypes as C
    lib, h = gb.lib, gb._capi.handle
    rng = np.random.default_rng(77)
    n = 60
    dense = rng.integers(0, 2, (n, n)).astype(bool)
    I, J = np.nonzero(dense)
    X = rng.integers(0, 2**32, len(I), dtype=np.uint64).astype(np.uint32)
    ux = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    A = gb.Matrix.from_arrays(I.astype(np.uint64), J.astype(np.uint64), X, n, n, gb.UINT32)
    u = gb.Vector.from_arrays(np.arange(n, dtype=np.uint64), ux, n, gb.UINT32)
    mon, sr = C.c_void_p(), C.c_void_p()
    assert lib.GrB_Monoid_new_UINT32(C.byref(mon), C.c_void_p(h("GrB_BOR_UINT32")), C.c_uint32(0)) == 0
    assert lib.GrB_Semiring_new(C.byref(sr), mon, C.c_void_p(h("GrB_BAND_UINT32"))) == 0
    w = gb.Vector.sparse(gb.UINT32, n)
    assert lib.GrB_mxv(w._h, None, None, sr, A._h, u._h, None) == 0
    gi, gx = w.to_arrays()
    M = np.zeros((n, n), np.uint32); M[I, J] = X
    exp = np.array([np.bitwise_or.reduce(M[i, dense[i]] & ux[dense[i]]) if dense[i].any() else 0 for i in range(n)], np.uint32)
    rows = np.flatnonzero(dense.any(axis=1))
    assert np.array_equal(gi.astype(np.int64), rows) and np.array_equal(gx, exp[rows])
    lib.GrB_Semiring_free(C.byref(sr)); lib.GrB_Monoid_free(C.byref(mon))
    # PLUS monoid with the POW multiplier: refused, with a reason
    Af = gb.Matrix.from_arrays(I.astype(np.uint64), J.astype(np.uint64), X.astype(np.float64), n, n, gb.FP64)
    uf = gb.Vector.from_arrays(np.arange(n, dtype=np.uint64), ux.astype(np.float64), n, gb.FP64)
    wf = gb.Vector.sparse(gb.FP64, n)
    sr2 = C.c_void_p()
    assert lib.GrB_Semiring_new(C.byref(sr2), C.c_void_p(h("GrB_PLUS_MONOID_FP64")), C.c_void_p(h("GxB_POW_FP64"))) == 0
    info = lib.GrB_mxv(wf._h, None, None, sr2, Af._h, uf._h, None)
    assert info == 5                                                # GrB_INVALID_VALUE: "not implemented in the MI355X backend: ..."
    msg = C.c_char_p()
    lib.GrB_Vector_error(C.byref(msg), wf._h)
    assert b"not implemented" in msg.value
    assert wf.nvals == 0
    lib.GrB_Semiring_free(C.byref(sr2))


def test_any_monoid_under_a_mask_takes_a_real_value(gpu):
    """Found by tools/fuzz_parity.py: the masked pull kernels reduce the lanes of a row with a tree that also saw the identity
    of the lanes without an entry — harmless for every monoid but ANY, which may keep either argument.  Rows longer than the
    lane-per-row prefix (8 entries) and than a row group, automatic kernel choice and the forced row-group kernel."""
    rng = np.random.default_rng(2)
    for typ in ("INT64", "FP32", "UINT8"):
        for method in (None, "rowgroup"):
            for vxm in (False, True):
                run_case(rng, typ, "ANY_PAIR", 120, 90, 0.5, 0.7, vxm=vxm, mask={"typ": "UINT8", "dens": 0.0, "comp": True}, method=method)
                run_case(rng, typ, "ANY_PAIR", 120, 90, 0.5, 0.7, vxm=vxm, mask={"typ": "BOOL", "dens": 0.6}, method=method)


def test_short_differential_fuzz_against_the_oracle(gpu):
    """Ten seconds of tools/fuzz_parity.py (random types / semirings / masks / accumulators / descriptors for mxv, vxm, mxm)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "--seconds", "10", "--seed", "11"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_kernel_x_extreme_shapes_at_scale(gpu):
    """Kernel X (panel pipeline + merge) on large matrices far from a graph: one dense row of 8 M entries (more sub-rows than a
    block of the row-major merge can hold: the per-panel merge takes over), a few dozen dense rows (wide rows added by a whole
    wave), millions of two-entry rows confined to one line of u (seven empty panels).  INT64 PLUS_TIMES, exact."""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    for nrows, ncols, kind in ((1, 1 << 23, "dense"), (48, 1 << 18, "dense"), (1 << 22, 8, "two")):
        if kind == "dense":
            rowptr = (torch.arange(nrows + 1, device=dev, dtype=torch.int64) * ncols).to(torch.int32)
            col = torch.arange(ncols, device=dev, dtype=torch.int32).repeat(nrows)
        else:
            rowptr = (torch.arange(nrows + 1, device=dev, dtype=torch.int64) * 2).to(torch.int32)
            c0 = torch.randint(0, 4, (nrows,), device=dev, generator=g, dtype=torch.int32)
            col = torch.stack([c0, c0 + 1 + torch.randint(0, 3, (nrows,), device=dev, generator=g, dtype=torch.int32)], 1).reshape(-1).contiguous()
        nnz = int(col.numel())
        vals = torch.randint(-3, 4, (nnz,), device=dev, generator=g, dtype=torch.int64)
        xs = torch.randint(-5, 6, (ncols,), device=dev, generator=g, dtype=torch.int64)
        A = gb.Matrix.from_csr(gb.INT64, nrows, ncols, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
        x = gb.Vector.from_dense_array((xs.data_ptr(), ncols), gb.INT64, device=True)
        os.environ["GRB_MI355X_SPMV"] = "xcd"
        try:
            w = A.mxv(x, semiring=gb.INT64.PLUS_TIMES)
            assert "k_spmv_xcd" in gb.last_kernel_plan(), gb.last_kernel_plan()
            w2 = A.mxv(x, semiring=gb.INT64.PLUS_TIMES)
        finally:
            os.environ.pop("GRB_MI355X_SPMV", None)
        gy, gp = w.to_dense_arrays()
        prod = vals * xs[col.to(torch.int64)]
        want = torch.zeros(nrows, dtype=torch.int64, device=dev).index_add_(0, torch.repeat_interleave(torch.arange(nrows, device=dev), (rowptr[1:] - rowptr[:-1]).to(torch.int64)), prod)
        assert gp.all() and np.array_equal(gy, want.cpu().numpy()), (nrows, ncols, kind)
        assert np.array_equal(w2.to_dense_arrays()[0], gy)


@pytest.mark.parametrize("typ", ["FP64", "FP32"])
@pytest.mark.parametrize("sr", ["MIN_DIV", "MAX_DIV", "MIN_RDIV"])
def test_nan_products_under_fp_min_max_agree_on_every_path(gpu, typ, sr, monkeypatch):
    """ONE NaN rule for a floating-point MIN / MAX monoid on every path (DESIGN.md §8, round-2 verdict): a NaN product (0/0 under DIV)
    is omitted by fmin / fmax, and an entry ALL of whose products are NaN is NaN — the oracle's first-product rule.  The same
    operands through every forced kernel (push SpMSpV and its atomics, row-block, row-group, the wave pipeline, the panel pipeline),
    with and without an accumulator and a mask, and as GrB_mxm through the masked LDS kernel, the two-pass hash and expand/sort/compress."""
    import helpers
    import test_mxm_gpu as TM
    # operands over {0, 1, 2}: one product in nine is 0/0, so entries whose products are ALL NaN are common
    monkeypatch.setattr(helpers, "rand_values", lambda r, t, n, small=True: r.integers(0, 3, n).astype(O.NP[t]))
    rng = np.random.default_rng(99)
    saw_nan = 0
    for method in (None, "adaptive", "rowgroup", "push", "wavepipe", "xcd"):
        for (nr, nc, dens, udens) in ((300, 280, 0.01, 0.5), (300, 280, 0.15, 0.5), (60, 9000, 0.6, 1.0), (4000, 30, 0.06, 1.0), (3000, 900, 0.002, 1.0)):
            for vxm in (False, True):
                for accum in (None, "MIN"):
                    if method in ("wavepipe", "xcd") and udens < 1.0:
                        continue
                    r = np.random.default_rng(int(rng.integers(1 << 30)))
                    run_case(r, typ, sr, nr, nc, dens, udens, vxm=vxm, accum=accum, method=method,
                             mask={"typ": "BOOL", "comp": True} if (method in (None, "rowgroup") and accum is None) else None)
    # at least one of the cases above must have produced an all-NaN entry, or the test shows nothing: count them through the oracle
    A = rand_matrix(rng, typ, 300, 280, 0.01); ui, ux = rand_vector(rng, typ, 280, 0.5)
    add, mul = sr.split("_")
    exp = O.mxv(O.col_vector(typ, 300), A, O.col_vector(typ, 280, ui, ux), add, mul, typ)
    saw_nan += int(np.isnan(exp.X).sum())
    assert saw_nan > 0
    for env in ({}, {"GRB_MI355X_SPGEMM": "esc"}, {"GRB_MI355X_MXM_ROWS": "1"}):
        for k in ("GRB_MI355X_SPGEMM", "GRB_MI355X_MXM_ROWS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for mask in (None, {"typ": "BOOL"}, {"typ": "BOOL", "comp": True}):
            TM.run_case(np.random.default_rng(5), typ, sr, 60, 50, 40, 0.05, 0.05, mask=mask, accum=None)
            TM.run_case(np.random.default_rng(7), typ, sr, 60, 50, 40, 0.3, 0.3, mask=mask, accum=None)
            TM.run_case(np.random.default_rng(6), typ, sr, 300, 20, 300, 0.5, 0.5, mask=mask, accum="MAX")


def test_reduce_bool_after_a_bool_product_reads_the_kernels_summary(gpu):
    """`while q.reduce_bool()` of the BFS loop (reference tests/test_bfs.py): the masked pull and the push kernels note whether
    they wrote a true value, and GrB_Vector_reduce_BOOL(LOR) reads that word instead of scanning q.  The answer must be the one a
    scan gives — with false-valued entries in the result (stored false values in A), an empty result, pull and push directions,
    after the vector changes (setElement, masked assign, clear), after a second product claimed the word, and for a dup."""
    rng = np.random.default_rng(77)
    n = 3000
    for dens_a, dens_u, a_true in ((0.004, 0.5, 0.5), (0.004, 0.001, 1.0), (0.004, 0.001, 0.0), (0.004, 0.3, 0.0), (0.02, 0.0005, 0.3)):
        nnz = int(n * n * dens_a); flat = np.sort(rng.choice(n * n, nnz, replace=False)); I, J = np.divmod(flat, n)
        X = rng.random(nnz) < a_true
        A = gb.Matrix.from_arrays(I.astype(np.uint64), J.astype(np.uint64), X, n, n, gb.BOOL)
        ui = np.sort(rng.choice(n, max(1, int(n * dens_u)), replace=False)).astype(np.uint64)
        u = gb.Vector.from_arrays(ui, np.ones(len(ui), np.bool_), n, gb.BOOL)
        mi = np.sort(rng.choice(n, n // 3, replace=False)).astype(np.uint64)
        seen = gb.Vector.from_arrays(mi, np.ones(len(mi), np.uint8), n, gb.UINT8)
        for mask, desc in ((seen, D.RC), (None, None), (seen, D.R)):
            q = gb.Vector.sparse(gb.BOOL, n)
            u.vxm(A, out=q, mask=mask, desc=desc)
            plan = gb.last_kernel_plan()
            got = q.reduce_bool()
            idx, val = vector_pairs(q)
            assert got == bool(np.any(val)), (plan, dens_a, dens_u, a_true)
            assert q.reduce_bool() == got                         # the cached answer
            d = q.dup(); assert d.reduce_bool() == got
            # a second product takes the word over: the first vector falls back to the scan
            q2 = gb.Vector.sparse(gb.BOOL, n); u.vxm(A, out=q2, mask=mask, desc=desc)
            q3 = gb.Vector.sparse(gb.BOOL, n); u.vxm(A, out=q3, mask=mask, desc=desc)
            assert q2.reduce_bool() == got and q3.reduce_bool() == got
            # changes after the product
            q2[5] = True; assert q2.reduce_bool() is True
            q3.assign_scalar(False, mask=seen); i3, v3 = vector_pairs(q3); assert q3.reduce_bool() == bool(np.any(v3))
            q.clear(); assert q.reduce_bool() is False
    # the loop itself, both directions taken along the way
    A = gb.Matrix.from_arrays(I.astype(np.uint64), J.astype(np.uint64), np.ones(nnz, np.bool_), n, n, gb.BOOL)
    v = gb.Vector.sparse(gb.UINT8, n); q = gb.Vector.sparse(gb.BOOL, n); q[0] = True
    level = 1; sizes = []
    while q.reduce_bool() and level <= n:
        sizes.append(q.nvals)
        v.assign_scalar(level, mask=q)
        v.vxm(A, mask=v, out=q, desc=D.RC)
        level += 1
    assert sum(sizes) == v.nvals and level > 2


def test_code_bytes_of_the_masked_pull_keep_explicit_zeros_exact(gpu, monkeypatch):
    """Round 6: the fused masked pull (`q<mask = v> = v lor.land A`, one-byte v) gathers ONE code byte per neighbour (bit 0 present, bit 1 present and not zero)
    instead of a presence byte and a value byte.  A level vector with EXPLICIT ZEROS — present entries whose value is false — must give the oracle's result,
    pattern included (a row whose only contributions are false has an entry, false): valued / structural / complemented masks, the code bytes left behind by
    the masked assign (`v[q] = level`) and rebuilt after another writer touched the vector, against the oracle and against the two-gather kernel."""
    rng = np.random.default_rng(77)
    n = 1 << 17
    key = np.unique(rng.integers(0, n * n, size=1300000, dtype=np.int64))
    I, J = np.divmod(key.astype(np.uint64), np.uint64(n))
    A_t = O.Tuples("BOOL", n, n, I, J, rng.random(len(key)) < 0.9)                                  # some stored `false` edges too
    A = to_matrix(A_t)
    empty = O.row_vector("BOOL", n)
    def oracle(vi, vx, struct, comp):
        u = O.row_vector("UINT8", n, vi, vx)
        return O.vxm(empty, u, A_t, "LOR", "LAND", "BOOL", mask=u, replace=True, mask_comp=comp, mask_struct=struct)
    for case in range(3):
        vi = np.sort(rng.choice(n, size=n // 3, replace=False)).astype(np.uint64)
        vx = rng.integers(0, 4, len(vi)).astype(np.uint8)                                              # a quarter of the entries are explicit zeros
        for struct, comp in ((False, True), (True, True), (False, False), (True, False)):
            flags = "R" + ("S" if struct else "") + ("C" if comp else "")
            want = oracle(vi, vx, struct, comp)
            got = {}
            for code in ("1", "0"):
                monkeypatch.setenv("GRB_MI355X_CODE_BYTES", code)
                v = to_vector("UINT8", n, vi, vx); q = gb.Vector.sparse(gb.BOOL, n)
                v.vxm(A, mask=v, out=q, desc=getattr(D, flags))
                plan = gb.last_kernel_plan()
                assert "mask=operand" in plan and ("code bytes" in plan) == (code == "1"), plan
                gi, gx = vector_pairs(q)
                assert np.array_equal(gi, want.J) and np.array_equal(gx.astype(bool), want.X.astype(bool)), (case, flags, code, plan)
                got[code] = plan
    # the producer: v[q] = level writes the code bytes; a later writer (apply in place) invalidates them and the pull rebuilds them
    monkeypatch.delenv("GRB_MI355X_CODE_BYTES", raising=False)
    vi = np.sort(rng.choice(n, size=n // 4, replace=False)).astype(np.uint64); vx = rng.integers(0, 3, len(vi)).astype(np.uint8)
    v = to_vector("UINT8", n, vi, vx)
    qi = np.sort(rng.choice(n, size=n // 5, replace=False)).astype(np.uint64); qx = rng.random(len(qi)) < 0.7
    qm = to_vector("BOOL", n, qi, qx)
    v.assign_scalar(7, mask=qm)
    dense = np.zeros(n, np.uint8); pres = np.zeros(n, bool); dense[vi.astype(int)] = vx; pres[vi.astype(int)] = True
    sel = qi[qx].astype(int); dense[sel] = 7; pres[sel] = True
    out = gb.Vector.sparse(gb.BOOL, n); v.vxm(A, mask=v, out=out, desc=D.RC)
    want = oracle(np.flatnonzero(pres).astype(np.uint64), dense[pres], False, True)
    gi, gx = vector_pairs(out); assert np.array_equal(gi, want.J) and np.array_equal(gx.astype(bool), want.X.astype(bool))
    v.apply(gb.UINT8.IDENTITY, out=v); v.assign_scalar(0, mask=qm, desc=D.S)                           # other writers: every position q holds becomes an explicit zero
    dense[qi.astype(int)] = 0; pres[qi.astype(int)] = True
    out2 = gb.Vector.sparse(gb.BOOL, n); v.vxm(A, mask=v, out=out2, desc=D.RC)
    want2 = oracle(np.flatnonzero(pres).astype(np.uint64), dense[pres], False, True)
    gi, gx = vector_pairs(out2); assert np.array_equal(gi, want2.J) and np.array_equal(gx.astype(bool), want2.X.astype(bool))


def test_min_plus_over_an_operand_with_holes_runs_the_full_operand_kernels(gpu):
    """The sweeps of the reference's shortest-path loop (`v<accum MIN> = v MIN_PLUS A`, demo/Intro-Prez.ipynb:1034-1045): the operand
    has no entry for vertices not reached yet.  grb_mxv.cpp fills the holes with a BIG value, runs a full-operand kernel and
    drops the sums made of fill values only — exact when the values are small against the type's range, otherwise the bitmap
    kernel runs as before.  Both directions of the decision, MIN and MAX, four types, rows fed by holes only, against the oracle."""
    rng = np.random.default_rng(31)
    n, nnz = 50000, 1300000
    key = np.unique(rng.integers(0, n * n, size=nnz, dtype=np.int64))
    I, J = np.divmod(key.astype(np.uint64), np.uint64(n))
    # the second half of the vertices never carries an entry of u: rows gathering only from there have no entry in the result
    for typ in ("INT64", "INT32", "FP64", "FP32"):
        for sr_name in ("MIN_PLUS", "MAX_PLUS"):
            for big_values in (False, True):
                dt = O.NP[typ]
                if typ.startswith("FP"):
                    X = (rng.integers(-40, 400, len(key)) / 8.0).astype(dt)
                    if big_values: X[7] = np.finfo(dt).max / 2                     # sums could overflow: the trick must not be used
                else:
                    X = rng.integers(-40, 400, len(key)).astype(dt)
                    if big_values: X[7] = np.iinfo(dt).max // 2
                A = O.Tuples(typ, n, n, I, J, X)
                ui = np.sort(rng.choice(n // 2, size=n // 5, replace=False)).astype(np.uint64)
                ux = (rng.integers(-100, 1000, len(ui)) / (8.0 if typ.startswith("FP") else 1)).astype(dt)
                wi = np.sort(rng.choice(n, size=n // 3, replace=False)).astype(np.uint64)
                wx = (rng.integers(-100, 1000, len(wi)) / (8.0 if typ.startswith("FP") else 1)).astype(dt)
                for accum in (None, sr_name.split("_")[0]):
                    gA, gu, gw = to_matrix(A), to_vector(typ, n, ui, ux), to_vector(typ, n, wi, wx)
                    gA.mxv(gu, semiring=getattr(TYPE[typ], sr_name), out=gw, accum=getattr(TYPE[typ], accum) if accum else None)
                    plan = gb.last_kernel_plan()
                    add, mul = sr_name.split("_")
                    exp = O.mxv(O.col_vector(typ, n, wi, wx), A, O.col_vector(typ, n, ui, ux), add, mul, typ, accum=accum, accum_type=typ)
                    gi, gx = vector_pairs(gw)
                    assert_same(typ, gi, gx, exp.I, exp.X, what=f"{typ}.{sr_name} accum={accum} big_values={big_values} plan={plan}")
                    if accum is None:
                        assert len(gi) < n                                          # some rows gather from holes only
                    if not big_values:
                        assert "k_spmv_wavepipe" in plan or "k_spmv_xcd" in plan, plan
                    else:
                        assert "k_spmv_wavepipe" not in plan and "k_spmv_xcd" not in plan, plan


def test_deterministic_mode_keeps_fp_sums_out_of_the_push_kernels(gpu, monkeypatch):
    """A sparse FP64 operand normally takes the push step (SpMSpV: one atomic per product, landing in any order).  With GRB_MI355X_DETERMINISTIC=1 a
    floating-point PLUS product pulls instead (fixed lane partition, fixed combination order): the same bits every time, the oracle's values to 1e-12;
    MIN / MAX monoids (any order gives the same bits) and integers keep the push step."""
    rng = np.random.default_rng(31)
    n = 60000
    A = rand_matrix(rng, "FP64", n, n, 0.0005, small=False)
    ui, ux = rand_vector(rng, "FP64", n, 0.002, small=False)
    M = to_matrix(A); u = to_vector("FP64", n, ui, ux)
    w = u.vxm(M, semiring=gb.FP64.PLUS_TIMES)
    assert "push" in gb.last_kernel_plan(), gb.last_kernel_plan()
    monkeypatch.setenv("GRB_MI355X_DETERMINISTIC", "1")
    ref = None
    for _ in range(4):
        w = u.vxm(M, semiring=gb.FP64.PLUS_TIMES)
        assert "push" not in gb.last_kernel_plan(), gb.last_kernel_plan()
        gi, gx = vector_pairs(w)
        cur = (np.asarray(gi).tobytes(), np.asarray(gx, np.float64).tobytes())
        if ref is None: ref = cur
        assert cur == ref
    gi, gx = vector_pairs(w)
    want = np.zeros(n)
    uval = np.zeros(n); uval[ui.astype(np.int64)] = ux; upres = np.zeros(n, bool); upres[ui.astype(np.int64)] = True
    sel = upres[A.I.astype(np.int64)]
    np.add.at(want, A.J.astype(np.int64)[sel], A.X[sel] * uval[A.I.astype(np.int64)[sel]])
    pres = np.zeros(n, bool); pres[A.J.astype(np.int64)[sel]] = True
    assert np.array_equal(np.asarray(gi, np.int64), np.nonzero(pres)[0]) and np.allclose(np.asarray(gx), want[pres], rtol=1e-12, atol=0)
    u.vxm(M, semiring=gb.FP64.MIN_PLUS)
    assert "push" in gb.last_kernel_plan(), gb.last_kernel_plan()


def test_summary_words_survive_the_wrap_of_their_24_bit_tags(gpu):
    """The product's host words carry a 24-bit tag; every 2^24 products the tags start over and the words are wiped (a word left unread 2^24 calls ago
    must not pass for the coming product's).  In a process of its own whose tags start eight below the wrap (GRB_MI355X_SUMMARY_TAG0): products of
    different sizes whose summaries are not all asked for, across the wrap — every `reduce_bool` is what a scan of the result gives."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import numpy as np, sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import pygraphblas_amd as gb
        from pygraphblas_amd import descriptor as D
        rng = np.random.default_rng(5)
        mats = []
        for n in (300000, 2000):                       # many workgroups / few workgroups
            nnz = n * 4; flat = np.unique(rng.integers(0, n * n, nnz, dtype=np.int64)); nnz = len(flat); I, J = [x.astype(np.uint64) for x in np.divmod(flat, n)]
            mats.append((n, gb.Matrix.from_arrays(I, J, np.ones(nnz, np.bool_), n, n, gb.BOOL)))
        for it in range(40):
            n, A = mats[0] if it %% 7 == 0 else mats[1]
            seen = gb.Vector.from_arrays(np.arange(0, n, 3, dtype=np.uint64), np.ones(len(range(0, n, 3)), np.uint8), n, gb.UINT8)
            ui = np.sort(rng.choice(n, 5 if it %% 3 else n // 2, replace=False)).astype(np.uint64)
            u = gb.Vector.from_arrays(ui, np.ones(len(ui), np.bool_), n, gb.BOOL)
            q = gb.Vector.sparse(gb.BOOL, n)
            u.vxm(A, out=q, mask=seen, desc=D.RC)
            if it %% 2:                                 # every other summary is never read
                got = q.reduce_bool()
                assert got == (q.nvals > 0), (it, got, q.nvals)
        print("ok")
    ''') % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GRB_MI355X_SUMMARY_TAG0=str(0xFFFFF8))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
