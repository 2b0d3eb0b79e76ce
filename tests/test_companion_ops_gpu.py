"""GPU tests of the O(n) / O(nnz) operations around the hot path (SURVEY.md §8f "next" rows 1-3): eWiseAdd/eWiseMult,
apply, select (tril/triu/offdiag/value tests), transpose, reduce to vector / scalar, assign_scalar, bulk CSR import/export,
typecasts — against numpy / scipy computed on the host.  Integer results bit-exact; FP on a 1/8 grid (exact)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as O
import pygraphblas_amd as gb
from pygraphblas_amd import descriptor as D
from helpers import TYPE, rand_matrix, rand_vector, to_matrix, to_vector

pytestmark = pytest.mark.gpu


def dense_m(t):
    d = np.zeros((t.nrows, t.ncols), dtype=t.X.dtype); p = np.zeros((t.nrows, t.ncols), bool)
    d[t.I.astype(int), t.J.astype(int)] = t.X; p[t.I.astype(int), t.J.astype(int)] = True
    return d, p


def got_m(m):
    I, J, X = m.to_arrays()
    d = np.zeros(m.shape, dtype=X.dtype); p = np.zeros(m.shape, bool)
    d[I.astype(int), J.astype(int)] = X; p[I.astype(int), J.astype(int)] = True
    return d, p


def got_v(v):
    return v.to_dense_arrays()


@pytest.mark.parametrize("typ", ["INT64", "FP64", "INT8", "UINT16", "FP32", "BOOL"])
def test_matrix_ewise_union_and_intersection(gpu, typ):
    rng = np.random.default_rng(1)
    A, B = rand_matrix(rng, typ, 40, 30, 0.3), rand_matrix(rng, typ, 40, 30, 0.3)
    da, pa = dense_m(A); db, pb = dense_m(B)
    T = TYPE[typ]
    add = T.LOR if typ == "BOOL" else T.PLUS
    mul = T.LAND if typ == "BOOL" else T.TIMES
    with np.errstate(over="ignore"):
        u = np.where(pa & pb, (da | db) if typ == "BOOL" else (da + db).astype(da.dtype), np.where(pa, da, db))
        i = (da & db) if typ == "BOOL" else (da * db).astype(da.dtype)
    g, p = got_m(to_matrix(A).eadd(to_matrix(B), add))
    assert np.array_equal(p, pa | pb) and np.array_equal(g[p], u[p])
    g, p = got_m(to_matrix(A).emult(to_matrix(B), mul))
    assert np.array_equal(p, pa & pb) and np.array_equal(g[p], i[p])
    # masked, complemented, with accum into an existing matrix
    Cm, M = rand_matrix(rng, typ, 40, 30, 0.3), rand_matrix(rng, "BOOL", 40, 30, 0.5)
    dc, pc = dense_m(Cm); dm, pm = dense_m(M)
    out = to_matrix(Cm)
    to_matrix(A).emult(to_matrix(B), mul, out=out, mask=to_matrix(M), accum=add, desc=D.C)
    allow = ~(pm & dm.astype(bool))
    tp = pa & pb
    with np.errstate(over="ignore"):
        z = np.where(pc & tp, (dc | i) if typ == "BOOL" else (dc + i).astype(dc.dtype), np.where(tp, i, dc))
    zp = pc | tp
    exp_p = np.where(allow, zp, pc); exp = np.where(allow, z, dc)
    g, p = got_m(out)
    assert np.array_equal(p, exp_p) and np.array_equal(g[p], exp[p])


def test_select_transpose_pattern_reduce_vector(gpu):
    rng = np.random.default_rng(2)
    A = rand_matrix(rng, "INT64", 50, 50, 0.2)
    d, p = dense_m(A); m = to_matrix(A)
    for name, keep in (("tril", np.tril(np.ones((50, 50), bool))), ("triu", np.triu(np.ones((50, 50), bool))), ("offdiag", ~np.eye(50, dtype=bool))):
        g, gp = got_m(getattr(m, name)())
        assert np.array_equal(gp, p & keep) and np.array_equal(g[gp], d[gp])
    g, gp = got_m(m.tril(-1)); assert np.array_equal(gp, p & np.tril(np.ones((50, 50), bool), -1))
    g, gp = got_m(m.select(">", 10)); assert np.array_equal(gp, p & (d > 10))
    g, gp = got_m(m.select("==0")); assert np.array_equal(gp, p & (d == 0))
    g, gp = got_m(m.nonzero()); assert np.array_equal(gp, p & (d != 0))
    g, gp = got_m(m.transpose()); assert np.array_equal(gp, p.T) and np.array_equal(g[gp], d.T[gp])
    g, gp = got_m(m.T.T); assert np.array_equal(gp, p) and np.array_equal(g[gp], d[gp])
    g, gp = got_m(m.pattern()); assert np.array_equal(gp, p) and g[gp].all()
    rv, rp = got_v(m.reduce_vector())
    assert np.array_equal(rp.astype(bool), p.any(1)) and np.array_equal(rv[rp != 0], np.where(p, d, 0).sum(1)[p.any(1)])
    rv, rp = got_v(m.reduce_vector(gb.INT64.MAX_MONOID, desc=D.T0))
    assert np.array_equal(rp.astype(bool), p.any(0)) and np.array_equal(rv[rp != 0], np.where(p, d, -2**62).max(0)[p.any(0)])
    assert m.reduce_int() == int(np.where(p, d, 0).sum()) and m.reduce_int(gb.INT64.MIN_MONOID) == int(d[p].min())
    assert gb.Matrix.sparse(gb.INT64, 5, 5).reduce_int() == 0


def test_one_pass_iseq_equals_the_composed_one(gpu):
    """`Vector.iseq` (pygraphblas/vector.py:188-235: size, nvals, eWiseMult(EQ) into BOOL, nvals, LAND-reduce) has a one-pass form in the library
    (`GrBX_Vector_iseq`, taken by the mirror for two vectors of one type and the type's own EQ): the same answers as the composed calls on equal vectors,
    a differing value, a differing pattern with the same entry count, NaN (differs from itself), -0.0 (equals 0.0), empty and full vectors, every type,
    a length that is not a multiple of eight, and operands with deferred work pending."""
    import ctypes as C
    rng = np.random.default_rng(17)
    def composed(a, b):
        if a.size != b.size or a.nvals != b.nvals: return False
        c = a.emult(b, a.type.EQ, cast=gb.BOOL)
        return c.nvals == a.nvals and bool(c.reduce_bool(gb.BOOL.LAND_MONOID))
    def fused(a, b):
        r = C.c_bool(False); info = gb.lib.GrBX_Vector_iseq(C.byref(r), a._h, b._h)
        assert info == 0, info
        return bool(r.value)
    for typ in ("BOOL", "INT8", "UINT16", "INT32", "INT64", "UINT64", "FP32", "FP64"):
        for n in (1, 7, 64, 1003, 70001):
            idx, vals = rand_vector(rng, typ, n, float(rng.choice([0.0, 0.3, 1.0])))
            u = to_vector(typ, n, idx, vals); v = to_vector(typ, n, idx, vals)
            assert fused(u, v) and composed(u, v) and u.iseq(v), (typ, n)
            if len(idx):
                k = int(rng.integers(len(idx))); v2 = vals.copy(); v2[k] = (not v2[k]) if typ == "BOOL" else v2[k] + 1
                w = to_vector(typ, n, idx, v2)
                assert not fused(u, w) and not composed(u, w) and not u.iseq(w), (typ, n, "value")
                if len(idx) < n:                                        # the same number of entries, one of them somewhere else
                    free = np.setdiff1d(np.arange(n, dtype=np.uint64), idx); i2 = idx.copy(); i2[k] = free[int(rng.integers(len(free)))]; o = np.argsort(i2)
                    x = to_vector(typ, n, i2[o], vals[o])
                    assert not fused(u, x) and not composed(u, x) and not u.iseq(x), (typ, n, "pattern")
    a = to_vector("FP64", 5, np.array([0, 2], np.uint64), np.array([np.nan, -0.0])); b = to_vector("FP64", 5, np.array([0, 2], np.uint64), np.array([np.nan, 0.0]))
    assert not fused(a, b) and not composed(a, b)
    c = to_vector("FP64", 5, np.array([2], np.uint64), np.array([-0.0])); d = to_vector("FP64", 5, np.array([2], np.uint64), np.array([0.0]))
    assert fused(c, d) and composed(c, d)
    e = gb.Vector.sparse(gb.INT64, 9); f = gb.Vector.sparse(gb.INT64, 9); g = gb.Vector.sparse(gb.INT64, 10)
    assert fused(e, f) and e.iseq(f) and not e.iseq(g)
    r = C.c_bool(); assert gb.lib.GrBX_Vector_iseq(C.byref(r), e._h, gb.Vector.sparse(gb.FP64, 9)._h) == 1           # GrB_NO_VALUE: two types — the mirror composes it
    assert not e.iseq(gb.Vector.sparse(gb.FP64, 10))
    # deferred work on an operand: p = x + y is queued (non-blocking mode) when it is compared
    n = 5000; ix, vx = rand_vector(rng, "FP64", n, 1.0); x = to_vector("FP64", n, ix, vx); y = to_vector("FP64", n, ix, vx)
    p = x.eadd(y, gb.FP64.PLUS); q = x.eadd(y, gb.FP64.PLUS)
    assert p.iseq(q) and not p.iseq(x)


def test_positional_unary_operators(gpu):
    """GxB_POSITIONI / POSITIONI1 / POSITIONJ / POSITIONJ1 (INT32, INT64; the reference lists them in pygraphblas/unaryop.py:55-63 and its notebooks use
    `A.apply(INT64.POSITIONJ)`, `A.positioni1()`): the result has the operand's pattern, an entry's value is its row / column index (0- or 1-based) in
    op(A) — whatever type the operand holds — and goes through the usual mask / accumulator / replace write-back.  A vector is an n x 1 column."""
    rng = np.random.default_rng(41)
    nr, nc = 70, 45
    ii, jj = np.meshgrid(np.arange(nr), np.arange(nc), indexing="ij")
    for styp in ("FP32", "INT64", "BOOL"):
        A = rand_matrix(rng, styp, nr, nc, 0.2); d, p = dense_m(A); m = to_matrix(A)
        for ztyp in ("INT64", "INT32"):
            Z = getattr(gb, ztyp)
            for name, exp in (("POSITIONI", ii), ("POSITIONI1", ii + 1), ("POSITIONJ", jj), ("POSITIONJ1", jj + 1)):
                out = gb.Matrix.sparse(Z, nr, nc); m.apply(getattr(Z, name), out=out)
                g, gp = got_m(out); assert np.array_equal(gp, p) and np.array_equal(g[gp], exp[gp]), (styp, ztyp, name)
            # the transposed operand: positions in A'
            out = gb.Matrix.sparse(Z, nc, nr); m.apply(Z.POSITIONI, out=out, desc=D.T0)
            g, gp = got_m(out); assert np.array_equal(gp, p.T) and np.array_equal(g[gp], jj.T[gp])
    # mask, accumulator, replace
    A = rand_matrix(rng, "INT64", nr, nc, 0.3); d, p = dense_m(A); m = to_matrix(A)
    Cm = rand_matrix(rng, "INT64", nr, nc, 0.3); cd, cp = dense_m(Cm)
    Mk = rand_matrix(rng, "BOOL", nr, nc, 0.5); md, mp = dense_m(Mk); allow = mp & (md != 0)
    out = to_matrix(Cm); m.apply(gb.INT64.POSITIONJ1, out=out, mask=to_matrix(Mk), accum=gb.INT64.PLUS)
    g, gp = got_m(out)
    ep = cp | (allow & p); e = np.where(allow & p, np.where(cp, cd, 0) + (jj + 1), cd)
    assert np.array_equal(gp, ep) and np.array_equal(g[gp], e[gp])
    out = to_matrix(Cm); m.apply(gb.INT64.POSITIONI, out=out, mask=to_matrix(Mk), desc=D.R)
    g, gp = got_m(out); assert np.array_equal(gp, allow & p) and np.array_equal(g[gp], ii[gp])
    # what the notebooks do with it: the column of every entry, reduced per row (demo/Louvain2.ipynb:52), one-based row labels (demo/Centrality.ipynb:653)
    S = gb.Matrix.from_arrays(np.arange(6, dtype=np.uint64), np.array([2, 0, 2, 5, 5, 1], np.uint64), np.ones(6, bool), 6, 6, gb.BOOL)
    lab, lp = got_v(S.cast(gb.INT64).apply(gb.INT64.POSITIONJ).reduce_vector())
    assert lp.all() and np.array_equal(lab, [2, 0, 2, 5, 5, 1])
    # vectors: the index, and column 0
    n = 300; idx, vals = rand_vector(rng, "FP64", n, 0.3); u = to_vector("FP64", n, idx, vals)
    for name, exp in (("POSITIONI", idx.astype(np.int64)), ("POSITIONI1", idx.astype(np.int64) + 1), ("POSITIONJ", np.zeros(len(idx), np.int64)), ("POSITIONJ1", np.ones(len(idx), np.int64))):
        w = gb.Vector.sparse(gb.INT64, n); u.apply(getattr(gb.INT64, name), out=w)
        g, gp = got_v(w); assert np.array_equal(np.flatnonzero(gp), idx.astype(np.int64)) and np.array_equal(g[gp != 0], exp), name


def test_matrix_apply_and_bound_scalars(gpu):
    rng = np.random.default_rng(3)
    A = rand_matrix(rng, "FP64", 30, 20, 0.3); d, p = dense_m(A); m = to_matrix(A)
    g, gp = got_m(m.apply(gb.FP64.AINV)); assert np.array_equal(gp, p) and np.array_equal(g[gp], -d[gp])
    g, gp = got_m(m.apply(gb.FP64.ABS)); assert np.array_equal(g[gp], np.abs(d[gp]))
    import ctypes as C
    out = gb.Matrix.sparse(gb.FP64, 30, 20)
    gb.base.check(gb.lib.GxB_Matrix_apply_BinaryOp2nd_FP64(out._h, None, None, C.c_void_p(gb.FP64.TIMES.get_op()), m._h, C.c_double(2.5), None), out)
    g, gp = got_m(out); assert np.array_equal(g[gp], d[gp] * 2.5)
    gb.base.check(gb.lib.GxB_Matrix_apply_BinaryOp1st_FP64(out._h, None, None, C.c_void_p(gb.FP64.MINUS.get_op()), C.c_double(1.0), m._h, None), out)
    g, gp = got_m(out); assert np.array_equal(g[gp], 1.0 - d[gp])
    # GrB_Matrix_assign_<T> over a row/column block, no mask
    I = np.array([1, 3], np.uint64); J = np.array([0, 2, 4], np.uint64)
    gb.base.check(gb.lib.GrB_Matrix_assign_FP64(m._h, None, None, C.c_double(9.0), I.ctypes.data_as(C.c_void_p), C.c_uint64(2), J.ctypes.data_as(C.c_void_p), C.c_uint64(3), None), m)
    e = d.copy(); ep = p.copy(); e[np.ix_([1, 3], [0, 2, 4])] = 9.0; ep[np.ix_([1, 3], [0, 2, 4])] = True
    g, gp = got_m(m); assert np.array_equal(gp, ep) and np.array_equal(g[gp], e[gp])


@pytest.mark.parametrize("typ", ["INT32", "FP32", "UINT8"])
def test_vector_ops(gpu, typ):
    rng = np.random.default_rng(4); n = 5000; T = TYPE[typ]
    ui, ux = rand_vector(rng, typ, n, 0.5); vi, vx = rand_vector(rng, typ, n, 0.5)
    u, v = to_vector(typ, n, ui, ux), to_vector(typ, n, vi, vx)
    du = np.zeros(n, ux.dtype); pu = np.zeros(n, bool); du[ui.astype(int)] = ux; pu[ui.astype(int)] = True
    dv = np.zeros(n, vx.dtype); pv = np.zeros(n, bool); dv[vi.astype(int)] = vx; pv[vi.astype(int)] = True
    with np.errstate(over="ignore"):
        s = (du + dv).astype(du.dtype); dmm = (du - dv).astype(du.dtype); pr = (du * dv).astype(du.dtype)
    g, p = got_v(u + v); assert np.array_equal(p.astype(bool), pu | pv) and np.array_equal(g[p != 0], np.where(pu & pv, s, np.where(pu, du, dv))[pu | pv])
    g, p = got_v(u - v); assert np.array_equal(g[p != 0], np.where(pu & pv, dmm, np.where(pu, du, dv))[pu | pv])
    g, p = got_v(u * v); assert np.array_equal(p.astype(bool), pu & pv) and np.array_equal(g[p != 0], pr[pu & pv])
    g, p = got_v(u.emult(v, T.MIN)); assert np.array_equal(g[p != 0], np.minimum(du, dv)[pu & pv])
    w = u.dup(); w -= v
    g, p = got_v(w); assert np.array_equal(g[p != 0], np.where(pu & pv, dmm, np.where(pu, du, dv))[pu | pv])
    g, p = got_v(abs(u)); assert np.array_equal(g[p != 0], np.abs(du[pu]))
    g, p = got_v(u.apply_second(T.PLUS, 3)); assert np.array_equal(g[p != 0], (du[pu] + du.dtype.type(3)).astype(du.dtype))
    # assign_scalar: all, masked (valued mask), with accum
    w = u.dup(); w.assign_scalar(7); g, p = got_v(w); assert p.all() and (g == 7).all()
    w = u.dup(); w.assign_scalar(5, mask=v)
    g, p = got_v(w); m = pv & (dv != 0)
    assert np.array_equal(p.astype(bool), pu | m) and np.array_equal(g[p != 0], np.where(m, 5, du)[pu | m])
    w = u.dup(); w.assign_scalar(2, accum=T.PLUS)
    g, p = got_v(w); assert p.all() and np.array_equal(g, np.where(pu, (du + du.dtype.type(2)).astype(du.dtype), 2))
    w = u.dup(); w.assign_scalar(1, index=[3, 4, 5])
    g, p = got_v(w); e = du.copy(); ep = pu.copy(); e[3:6] = 1; ep[3:6] = True
    assert np.array_equal(p.astype(bool), ep) and np.array_equal(g[p != 0], e[ep])
    # reductions
    if typ != "UINT8":
        assert u.reduce_int() == int(du[pu].astype(np.int64).sum()) if typ == "INT32" else abs(u.reduce_float() - float(du[pu].astype(np.float64).sum())) < 1e-3
    assert u.reduce_bool() == bool((du[pu] != 0).any())
    assert gb.Vector.sparse(T, 10).reduce_bool() is False


def test_entry_counts_follow_the_ops(gpu):
    """Vector ops carry known entry counts into their results (so the next product need not count on the device): the
    counts they carry must be the true ones."""
    rng = np.random.default_rng(11); n = 4000
    fi = np.arange(n, dtype=np.uint64); fx = rng.random(n) + 0.5
    si, sx = rand_vector(rng, "FP64", n, 0.3)
    full, full2, sp = to_vector("FP64", n, fi, fx), to_vector("FP64", n, fi, fx[::-1].copy()), to_vector("FP64", n, si, sx)
    for v in (full, full2, sp): got_v(v)                              # on the device, counts known
    def cnt(v):
        g, p = got_v(v); return int((p != 0).sum())
    for expr, expect in [(lambda: full * full2, n), (lambda: full + sp, n), (lambda: sp + full, n), (lambda: full * sp, len(si)),
                         (lambda: sp * sp, len(si)), (lambda: sp + sp, len(si)), (lambda: abs(full), n), (lambda: abs(sp), len(si)),
                         (lambda: full.apply_second(gb.FP64.PLUS, 1.0), n)]:
        r = expr(); assert r.nvals == expect == cnt(r)
    w = sp.dup(); w.assign_scalar(3.0); assert w.nvals == n == cnt(w)
    w = sp.dup(); w.assign_scalar(3.0, mask=sp); assert w.nvals == cnt(w)
    w = full.dup(); w.assign_scalar(2.0, accum=gb.FP64.PLUS); assert w.nvals == n == cnt(w)
    w = full.dup(); sp.apply(gb.FP64.ABS, out=w, accum=gb.FP64.PLUS); assert w.nvals == n == cnt(w)        # union with a full w stays full
    w = sp.dup(); full.apply(gb.FP64.ABS, out=w, accum=gb.FP64.PLUS); assert w.nvals == n == cnt(w)        # ... or with a full T
    w = sp.dup(); sp.apply(gb.FP64.ABS, out=w, accum=gb.FP64.PLUS); assert w.nvals == len(si) == cnt(w)


def test_math_library_operators(gpu):
    """The O(n) kernels exist in two variants (with / without the operators that call the math library): exercise the heavy one."""
    rng = np.random.default_rng(9); n = 3000
    ui, ux = rand_vector(rng, "FP64", n, 0.6); vi, vx = rand_vector(rng, "FP64", n, 0.6)
    ux = np.abs(ux) + 0.5; vx = np.abs(vx) * 0.25
    u, v = to_vector("FP64", n, ui, ux), to_vector("FP64", n, vi, vx)
    du = np.zeros(n); pu = np.zeros(n, bool); du[ui.astype(int)] = ux; pu[ui.astype(int)] = True
    dv = np.zeros(n); pv = np.zeros(n, bool); dv[vi.astype(int)] = vx; pv[vi.astype(int)] = True
    g, p = got_v(u.emult(v, gb.FP64.POW)); assert np.array_equal(p.astype(bool), pu & pv) and np.allclose(g[p != 0], np.power(du, dv)[pu & pv], rtol=1e-12)
    g, p = got_v(u.eadd(v, gb.FP64.ATAN2)); both = pu & pv
    assert np.allclose(g[both], np.arctan2(du, dv)[both], rtol=1e-12) and np.array_equal(g[pu & ~pv], du[pu & ~pv])
    g, p = got_v(u.apply(gb.FP64.EXP)); assert np.allclose(g[p != 0], np.exp(du[pu]), rtol=1e-12)
    g, p = got_v(u.apply(gb.FP64.LOG)); assert np.allclose(g[p != 0], np.log(du[pu]), rtol=1e-12)
    w = u.dup(); w.assign_scalar(2.0, accum=gb.FP64.POW); g, p = got_v(w)
    assert p.all() and np.allclose(g, np.where(pu, du ** 2.0, 2.0), rtol=1e-12)


@pytest.mark.parametrize("n", [1, 15, 16, 17, 1000, 4099, 70001])
def test_bool_reduce_lor_land(gpu, n):
    """BOOL vectors reduce with LOR (the BFS loop condition) and LAND in one counting kernel: present/absent x true/false."""
    rng = np.random.default_rng(n)
    for dens, ptrue in [(0.5, 0.5), (0.3, 0.0), (0.3, 1.0), (1.0, 0.5), (1.0, 1.0), (0.02, 0.5)]:
        idx = np.flatnonzero(rng.random(n) < dens).astype(np.uint64)
        if len(idx) == 0: idx = np.array([n - 1], np.uint64)
        val = rng.random(len(idx)) < ptrue
        u = to_vector("BOOL", n, idx, val)
        d, _ = got_v(u)                      # moves it to the device
        assert u.reduce_bool() == bool(val.any()), (n, dens, ptrue)
        assert u.reduce_bool(mon=gb.BOOL.LAND_MONOID) == bool(val.all()), (n, dens, ptrue)


def test_bulk_csr_roundtrip_and_scipy(gpu):
    rng = np.random.default_rng(6)
    S = sp.random(300, 200, density=0.05, format="csr", random_state=7, dtype=np.float64)
    S.sort_indices()
    m = gb.Matrix.from_scipy_sparse(S)
    assert m.nvals == S.nnz and m.type is gb.FP64
    rp, ci, x = m.to_csr()
    assert np.array_equal(rp, S.indptr.astype(np.uint32)) and np.array_equal(ci, S.indices.astype(np.uint32)) and np.array_equal(x, S.data)
    x0 = rng.random(200)
    y = m.mxv(gb.Vector.from_dense_array(x0, gb.FP64), semiring=gb.FP64.PLUS_TIMES)
    yv, yp = y.to_dense_arrays()
    assert np.allclose(yv[yp != 0], (S @ x0)[yp != 0], rtol=1e-12) and np.array_equal(yp.astype(bool), np.diff(S.indptr) > 0)
    assert (m.to_scipy_sparse() != S).nnz == 0
    # build with tuples given in arbitrary order == import of the sorted CSR
    perm = rng.permutation(S.nnz); coo = S.tocoo()
    m2 = gb.Matrix.from_arrays(coo.row[perm].astype(np.uint64), coo.col[perm].astype(np.uint64), coo.data[perm], 300, 200, gb.FP64)
    assert m2.iseq(m)


def test_entry_parallel_companions_on_a_power_law_graph(gpu):
    """select / eWise / masked write-back on R-MAT-15 (hub rows of thousands of entries, long runs of empty rows) against scipy:
    the entry-parallel kernels (scan, binary search, one stable merge) must not depend on how the entries split into rows."""
    from pygraphblas_amd import rmat
    scale = 15; n = 1 << scale
    rp, col = rmat.csr_numpy(scale, symmetric=True, drop_self_loops=True)
    rng = np.random.default_rng(3)
    av = rng.integers(1, 9, len(col)).astype(np.int64)
    A = gb.Matrix.from_csr(gb.INT64, n, n, rp, col, av)
    SA = sp.csr_matrix((av, col.astype(np.int64), rp.astype(np.int64)), shape=(n, n))
    rp2, col2 = rmat.csr_numpy(scale, seed=7)                                 # another graph, not symmetric
    bv = rng.integers(1, 9, len(col2)).astype(np.int64)
    B = gb.Matrix.from_csr(gb.INT64, n, n, rp2, col2, bv)
    SB = sp.csr_matrix((bv, col2.astype(np.int64), rp2.astype(np.int64)), shape=(n, n))

    def same(M, S):
        S = S.tocsr(); S.sort_indices()
        grp, gci, gx = M.to_csr()
        assert np.array_equal(grp.astype(np.int64), S.indptr) and np.array_equal(gci.astype(np.int64), S.indices) and np.array_equal(gx, S.data)
    same(A.tril(), sp.tril(SA)); same(A.triu(1), sp.triu(SA, 1)); same(A.offdiag(), SA - sp.diags(SA.diagonal()))
    same(A.select(">", 4), SA.multiply(SA > 4))
    same(A.eadd(B, gb.INT64.PLUS), SA + SB)
    PA = (SA != 0).astype(np.int64); PB = (SB != 0).astype(np.int64)
    same(A.emult(B, gb.INT64.TIMES), SA.multiply(SB))
    # C<L> = accum(C, B): entries of C outside the mask stay, inside they take C + B (or B where C has none)
    Lm = A.tril(); SL = (sp.tril(SA) != 0).astype(np.int64)
    C = A.dup()
    B.apply(gb.INT64.IDENTITY, out=C, mask=Lm, accum=gb.INT64.PLUS)
    want = SA + SB.multiply(SL)
    same(C, want)
    # ... and with replace: everything outside the mask goes
    C = A.dup()
    B.apply(gb.INT64.IDENTITY, out=C, mask=Lm, accum=gb.INT64.PLUS, desc=D.R)
    same(C, (SA + SB).multiply(SL))


def test_companions_fuzzed_against_a_model(gpu):
    """Six seconds of tools/fuzz_companions.py: eWise, apply, bound apply, select, transpose, reduce to a vector and scalar assign
    with random masks (valued / structural / complemented), accumulators, replace and transposed inputs against a Python model
    of the GraphBLAS rules (1.6e5 cases were clean when it was written)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_companions.py"), "--seconds", "6", "--seed", "5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz companions ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("tname,fill", [("FP32", 2.5), ("INT64", -7), ("BOOL", True), ("UINT8", 200)])
def test_dense_matrix_is_filled_on_the_device(gb, gpu, tname, fill):
    """`Matrix.dense(T, ns, n, fill)` / `M[:, :] = x` over every position (the ns x n batches of gap/bcmark.py:19-20, 48): one fill
    kernel in HBM; rows, columns and values of the resulting CSR, and the degenerate one-column / one-row shapes."""
    typ = getattr(gb, tname)
    for nr, nc in ((4, 100_003), (1, 70_000), (3000, 1), (37, 41)):
        m = gb.Matrix.dense(typ, nr, nc, fill=fill)
        assert m.nvals == nr * nc
        I, J, X = m.to_arrays()
        assert np.array_equal(I, np.repeat(np.arange(nr, dtype=np.uint64), nc))
        assert np.array_equal(J, np.tile(np.arange(nc, dtype=np.uint64), nr))
        assert np.all(X == typ._np(fill))
        rp, ci, vals = m.to_csr()
        assert np.array_equal(rp, np.arange(nr + 1, dtype=np.uint32) * np.uint32(nc))


def test_column_reduce_and_element_access_of_a_large_device_matrix(gpu):
    """The tail of gap/bcmark.py on an ns x n batch: `bcu.reduce_vector(accum=PLUS, out=cent, desc=T0)` reduces the columns of a
    by-row matrix without building its transpose (atomics per column), and `paths[i, source] = 1` / `paths[i, j]` on a matrix that
    lives in HBM only look the entry up on the device instead of bringing 10^7 tuples to the host mirror."""
    rng = np.random.default_rng(3)
    ns, n = 4, 300000
    nnz = 1200000
    key = np.unique(rng.integers(0, ns * n, size=nnz, dtype=np.int64))
    I, J = np.divmod(key.astype(np.uint64), np.uint64(n))
    for typ, acc in (("FP32", "PLUS"), ("INT64", "MIN"), ("FP64", "MAX")):
        X = rng.integers(-20, 50, len(key)).astype(O.NP[typ])
        M = gb.Matrix.from_arrays(I, J, X, ns, n, TYPE[typ])
        w = gb.Vector.from_arrays(np.arange(n, dtype=np.uint64), np.full(n, 3, O.NP[typ]), n, TYPE[typ])
        M.reduce_vector(getattr(TYPE[typ], acc + "_MONOID"), out=w, accum=getattr(TYPE[typ], "PLUS"), desc=D.T0)
        exp = np.full(n, 3, np.float64)
        red = np.zeros(n); has = np.zeros(n, bool)
        Xf = X.astype(np.float64); Ji = J.astype(np.int64)
        if acc == "PLUS": np.add.at(red, Ji, Xf)
        elif acc == "MIN": red[:] = np.inf; np.minimum.at(red, Ji, Xf)
        else: red[:] = -np.inf; np.maximum.at(red, Ji, Xf)
        has[Ji] = True
        exp[has] += red[has]
        wi, wx = w.to_arrays()
        assert len(wi) == n and np.array_equal(np.asarray(wx, np.float64), exp), (typ, acc)
    # element access on the device: a dense batch made in HBM
    P = gb.Matrix.dense(gb.FP32, ns, n, 0)
    P[2, 12345] = 1
    P[3, 0] = 7.5
    assert P[2, 12345] == 1 and P[3, 0] == 7.5 and P[0, 5] == 0 and P.nvals == ns * n
    r = gb.Vector.sparse(gb.FP32, n); P.reduce_vector(gb.FP32.PLUS_MONOID, out=r, desc=D.T0)
    assert r[12345] == 1 and r[0] == 7.5 and r[7] == 0                      # single elements of a vector that lives in HBM only
    for k in range(40):                                                          # ... more than a few dozen in a row: the host mirror takes over
        assert r[k + 1] == 0
    ri, rx = r.to_arrays()
    assert rx[12345] == 1 and rx[0] == 7.5 and rx.sum() == 8.5
    sv = gb.Vector.from_arrays(np.array([3, 70000], np.uint64), np.array([1.5, 2.5], np.float32), n, gb.FP32)
    sv2 = sv.apply(gb.FP32.AINV)                                                 # a device-only result with two entries
    assert sv2[70000] == -2.5 and sv2.get(5) is None and sv2.nvals == 2
    S = gb.Matrix.from_arrays(I, J, np.ones(len(key), np.float32), ns, n, gb.FP32)
    S2 = S.apply(gb.FP32.AINV)                                  # a device-only result
    i0, j0 = int(I[1000]), int(J[1000])
    assert S2[i0, j0] == -1
    S2[i0, j0] = 5                                              # an existing entry: in place
    assert S2[i0, j0] == 5 and S2.nvals == len(key)
    free = next(c for c in range(n) if not ((I == 0) & (J == c)).any())
    S2[0, free] = 9                                             # a new entry: the host mirror takes over
    assert S2[0, free] == 9 and S2.nvals == len(key) + 1 and S2[i0, j0] == 5


def test_column_reduce_of_a_few_long_fp_rows_does_not_depend_on_a_cached_transpose(gpu):
    """`reduce_vector(desc=T0)` with an FP PLUS monoid over a batch of <= 64 long rows (the last statement of gap/bcmark.py:66): one fixed-order algorithm
    whatever the matrix has cached — the same call returned other bits after an earlier product had built the transpose (round-4 advice)."""
    rng = np.random.default_rng(31)
    nr, nc = 4, 1 << 19
    keep = rng.random((nr, nc)) < 0.6
    I, J = np.nonzero(keep)
    X = rng.random(len(I)).astype(np.float32)
    assert len(I) >= 1 << 20
    A = gb.Matrix.from_arrays(I.astype(np.uint64), J.astype(np.uint64), X, nr, nc, gb.FP32)
    r1 = A.reduce_vector(gb.FP32.PLUS_MONOID, out=gb.Vector.sparse(gb.FP32, nc), desc=D.T0).to_arrays()
    u = gb.Vector.from_dense_array(np.ones(nr, np.float32), gb.FP32)
    A.mxv(u, semiring=gb.FP32.PLUS_TIMES, desc=D.T0)                       # builds and caches the transpose
    r2 = A.reduce_vector(gb.FP32.PLUS_MONOID, out=gb.Vector.sparse(gb.FP32, nc), desc=D.T0).to_arrays()
    assert np.array_equal(r1[0], r2[0]) and np.array_equal(r1[1].view(np.uint32), r2[1].view(np.uint32))
    want = np.zeros(nc, np.float64); np.add.at(want, J, X.astype(np.float64))
    assert np.allclose(r1[1], want[r1[0].astype(np.int64)], rtol=1e-5)
