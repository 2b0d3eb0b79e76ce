"""GPU parity of the hot path on hypersparse containers (GxB_INDEX_MAX = 2^60 dimensions, the reference's default:
pygraphblas/matrix.py:167-170; demo/Intro-Prez.ipynb "pass no dimension to go hypersparse" runs its BFS and shortest-path
loops on such a matrix).  grb_hyper.cpp relabels the indices that occur, runs the ordinary HIP operation and maps the
result back — so the reference here is the oracle on the SAME operands with their indices compacted by numpy, and the
results must agree entry for entry after mapping back (bit-exact: values on the 1/8 grid)."""
import zlib

import numpy as np
import pytest

from oracle import oracle as O
import pygraphblas_amd as gb
from pygraphblas_amd import descriptor as D
from helpers import TYPE, rand_matrix, rand_values

pytestmark = pytest.mark.gpu
IMAX = 1 << 60


def pool(rng, k):
    """k distinct indices spread over [0, 2^60): a few small ones, a few at the very top, the rest anywhere."""
    p = set(int(x) for x in rng.integers(0, IMAX, k, dtype=np.uint64))
    p |= {0, 1, IMAX - 1, IMAX - 2, (1 << 32) - 1, 1 << 32}
    return np.array(sorted(p), np.uint64)


def hyper_matrix(t, rp, cp):
    """oracle tuples over pool positions -> product matrix over the huge index space"""
    return gb.Matrix.from_arrays(rp[t.I.astype(np.int64)], cp[t.J.astype(np.int64)], t.X, IMAX, IMAX, TYPE[t.typ])


def back(m, rp, cp):
    I, J, X = m.to_arrays()
    return np.searchsorted(rp, I).astype(np.uint64), np.searchsorted(cp, J).astype(np.uint64), X


def same(got, exp, what):
    gi, gj, gx = got
    assert np.array_equal(gi, exp.I) and np.array_equal(gj, exp.J), f"pattern differs: {what} [{gb.last_kernel_plan()}]"
    assert np.array_equal(gx, exp.X), f"values differ: {what}: {gx[:8]} vs {exp.X[:8]}"


@pytest.mark.parametrize("typ,sr", [("INT64", "PLUS_TIMES"), ("FP64", "MIN_PLUS"), ("BOOL", "LOR_LAND"), ("FP32", "PLUS_SECOND")])
def test_mxm_hypersparse(gpu, typ, sr):
    rng = np.random.default_rng(zlib.crc32(f"{typ}.{sr}".encode()))
    add, mul = sr.split("_")
    rp, kp, cp = pool(rng, 40), pool(rng, 50), pool(rng, 30)
    for mask, accum, replace, ta, tb in ((None, None, False, False, False), ({"comp": True}, "PLUS" if typ != "BOOL" else "LOR", True, False, False),
                                          ({"struct": True}, None, False, True, False), (None, None, False, False, True)):
        A = rand_matrix(rng, typ, *((len(kp), len(rp)) if ta else (len(rp), len(kp))), 0.1)
        B = rand_matrix(rng, typ, *((len(cp), len(kp)) if tb else (len(kp), len(cp))), 0.1)
        Cm = rand_matrix(rng, typ, len(rp), len(cp), 0.1)
        M = rand_matrix(rng, "BOOL", len(rp), len(cp), 0.3) if mask else None
        gA = hyper_matrix(A, *((kp, rp) if ta else (rp, kp))); gB = hyper_matrix(B, *((cp, kp) if tb else (kp, cp)))
        gC = hyper_matrix(Cm, rp, cp); gM = hyper_matrix(M, rp, cp) if mask else None
        flags = "".join(["R" if replace else "", "S" if mask and mask.get("struct") else "", "C" if mask and mask.get("comp") else "", "T0" if ta else "", "T1" if tb else ""])
        gA.mxm(gB, semiring=getattr(TYPE[typ], sr), out=gC, mask=gM, accum=getattr(TYPE[typ], accum) if accum else None, desc=getattr(D, flags) if flags else None)
        assert gb.last_kernel_plan().startswith("hypersparse<"), gb.last_kernel_plan()
        exp = O.mxm(Cm, A, B, add, mul, typ, mask=M, accum=accum, accum_type=typ, replace=replace, mask_comp=bool(mask and mask.get("comp")),
                    mask_struct=bool(mask and mask.get("struct")), tran_a=ta, tran_b=tb)
        assert gC.nrows == IMAX and gC.ncols == IMAX
        same(back(gC, rp, cp), exp, f"{typ}.{sr} mask={mask} accum={accum} R={replace} ta={ta} tb={tb}")


@pytest.mark.parametrize("vxm", [False, True])
@pytest.mark.parametrize("typ,sr", [("INT64", "PLUS_TIMES"), ("FP64", "MIN_PLUS"), ("BOOL", "ANY_PAIR")])
def test_mxv_vxm_hypersparse(gpu, typ, sr, vxm):
    rng = np.random.default_rng(zlib.crc32(f"{typ}.{sr}.{vxm}".encode()))
    add, mul = sr.split("_")
    rp, cp = pool(rng, 60), pool(rng, 45)
    for mask, accum, replace, tr in ((None, None, False, False), ({"comp": True}, None, True, False), ({}, "MIN" if typ != "BOOL" else "LOR", False, True)):
        A = rand_matrix(rng, typ, len(rp), len(cp), 0.1)
        # op(A) as the product uses it: mxv w = op(A) u, vxm w = u op(A)
        orows, ocols = (cp, rp) if tr else (rp, cp)
        inp, outp = (orows, ocols) if vxm else (ocols, orows)
        ui = np.sort(rng.choice(len(inp), size=len(inp) // 3, replace=False)).astype(np.uint64); ux = rand_values(rng, typ, len(ui))
        wi = np.sort(rng.choice(len(outp), size=len(outp) // 4, replace=False)).astype(np.uint64); wx = rand_values(rng, typ, len(wi))
        mi = np.sort(rng.choice(len(outp), size=len(outp) // 2, replace=False)).astype(np.uint64) if mask is not None else None
        gA = hyper_matrix(A, rp, cp)
        gu = gb.Vector.from_arrays(inp[ui.astype(np.int64)], ux, IMAX, TYPE[typ]); gw = gb.Vector.from_arrays(outp[wi.astype(np.int64)], wx, IMAX, TYPE[typ])
        gm = gb.Vector.from_arrays(outp[mi.astype(np.int64)], np.ones(len(mi), np.bool_), IMAX, gb.BOOL) if mask is not None else None
        flags = "".join(["R" if replace else "", "C" if mask and mask.get("comp") else "", ("T1" if vxm else "T0") if tr else ""])
        kw = dict(semiring=getattr(TYPE[typ], sr), out=gw, mask=gm, accum=getattr(TYPE[typ], accum) if accum else None, desc=getattr(D, flags) if flags else None)
        (gu.vxm(gA, **kw) if vxm else gA.mxv(gu, **kw))
        assert gb.last_kernel_plan().startswith("hypersparse<"), gb.last_kernel_plan()
        w0 = (wi, wx); u0 = (ui, ux); m0 = (mi, np.ones(len(mi), np.bool_)) if mask is not None else None
        okw = dict(mask=O.col_vector("BOOL", len(outp), *m0) if (m0 and not vxm) else (O.row_vector("BOOL", len(outp), *m0) if m0 else None), accum=accum, accum_type=typ,
                   replace=replace, mask_comp=bool(mask and mask.get("comp")))
        if vxm:
            exp = O.vxm(O.row_vector(typ, len(outp), *w0), O.row_vector(typ, len(inp), *u0), A, add, mul, typ, tran_a=tr, **okw)
            ei, ex = exp.J, exp.X
        else:
            exp = O.mxv(O.col_vector(typ, len(outp), *w0), A, O.col_vector(typ, len(inp), *u0), add, mul, typ, tran_a=tr, **okw)
            ei, ex = exp.I, exp.X
        I, X = gw.to_arrays()
        assert gw.size == IMAX
        assert np.array_equal(np.searchsorted(outp, I).astype(np.uint64), ei), f"pattern {typ}.{sr} vxm={vxm} mask={mask} tr={tr} [{gb.last_kernel_plan()}]"
        assert np.array_equal(X, ex), f"values {typ}.{sr} vxm={vxm} mask={mask} tr={tr}: {X[:8]} vs {ex[:8]}"


def test_ewise_and_loops_hypersparse(gpu):
    rng = np.random.default_rng(5)
    p = pool(rng, 80)
    ui = np.sort(rng.choice(len(p), 30, replace=False)); vi = np.sort(rng.choice(len(p), 30, replace=False))
    ux, vx = rand_values(rng, "INT64", 30), rand_values(rng, "INT64", 30)
    gu, gv = gb.Vector.from_arrays(p[ui], ux, IMAX, gb.INT64), gb.Vector.from_arrays(p[vi], vx, IMAX, gb.INT64)
    du, dv = dict(zip(ui.tolist(), ux.tolist())), dict(zip(vi.tolist(), vx.tolist()))
    add = gu.eadd(gv, gb.INT64.PLUS); I, X = add.to_arrays()
    want = {k: du.get(k, 0) + dv.get(k, 0) for k in set(du) | set(dv)}
    assert dict(zip(np.searchsorted(p, I).tolist(), X.tolist())) == want and add.size == IMAX
    mul = gu.emult(gv, gb.INT64.TIMES); I, X = mul.to_arrays()
    assert dict(zip(np.searchsorted(p, I).tolist(), X.tolist())) == {k: du[k] * dv[k] for k in set(du) & set(dv)}
    assert gu.iseq(gu.dup()) and not gu.iseq(gv)
    # matrices
    A = rand_matrix(rng, "FP64", len(p), len(p), 0.02); B = rand_matrix(rng, "FP64", len(p), len(p), 0.02)
    gA, gB = hyper_matrix(A, p, p), hyper_matrix(B, p, p)
    S = gA.eadd(gB, gb.FP64.PLUS)
    import scipy.sparse as sp
    want = (sp.csr_matrix((A.X, (A.I.astype(np.int64), A.J.astype(np.int64))), shape=(len(p),) * 2) + sp.csr_matrix((B.X, (B.I.astype(np.int64), B.J.astype(np.int64))), shape=(len(p),) * 2)).tocoo()
    gi, gj, gx = back(S, p, p)
    got = sp.csr_matrix((gx, (gi.astype(np.int64), gj.astype(np.int64))), shape=(len(p),) * 2)
    assert (abs(got - want.tocsr()) > 0).nnz == 0 and S.nvals == len(set(zip(A.I.tolist(), A.J.tolist())) | set(zip(B.I.tolist(), B.J.tolist())))
    # the reference's BFS loop on a hypersparse graph (demo/Intro-Prez.ipynb cells 6-7): a path 0 -> 1 -> 5 -> 2^60 - 1
    edges = [(0, 1), (1, 5), (5, IMAX - 1), (0, 5), (7, 0)]
    G = gb.Matrix.from_lists([e[0] for e in edges], [e[1] for e in edges], [True] * len(edges), IMAX, IMAX, gb.BOOL)
    v = gb.Vector.sparse(gb.UINT8, IMAX); q = gb.Vector.sparse(gb.BOOL, IMAX); q[0] = True
    level = 1
    while q.nvals and level < 10:
        v.assign_scalar(level, mask=q)
        v.vxm(G, mask=v, semiring=gb.BOOL.ANY_PAIR, desc=D.RC, out=q)
        level += 1
    I, X = v.to_arrays()
    assert dict(zip(I.tolist(), X.tolist())) == {0: 1, 1: 2, 5: 2, IMAX - 1: 3}
