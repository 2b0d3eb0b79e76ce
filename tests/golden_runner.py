"""Run one case of tests/golden/reference_vectors.json through the oracle or through the product."""
import json
import os

import numpy as np

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
        return json.load(f)


def _np(typ, x):
    return np.asarray(x, dtype=O.NP[typ])


def flags(case):
    d = case.get("desc", "") or ""
    core = d.replace("T0", "").replace("T1", "")
    return dict(replace="R" in core, mask_struct="S" in core, mask_comp="C" in core, t0="T0" in d, t1="T1" in d)


def run_oracle(case):
    typ = case["type"]
    add, mul = case["semiring"].split("_")
    f = flags(case)
    et = case.get("expect_type", typ)
    at = case.get("A_type", "INT64")
    A = O.Tuples(at, case["A"][3], case["A"][4], case["A"][0], case["A"][1], _np(at, case["A"][2]))
    kw = dict(accum=case.get("accum"), accum_type=et, replace=f["replace"], mask_comp=f["mask_comp"], mask_struct=f["mask_struct"])
    if case["op"] == "mxm":
        bt = case.get("B_type", "INT64")
        B = O.Tuples(bt, case["B"][3], case["B"][4], case["B"][0], case["B"][1], _np(bt, case["B"][2]))
        nr = A.ncols if f["t0"] else A.nrows
        nc = B.nrows if f["t1"] else B.ncols
        Cm = O.Tuples(et, nr, nc, *(case["C"][:2] + [_np(et, case["C"][2])] if "C" in case else ([], [], [])))
        r = O.mxm(Cm, A, B, add, mul, typ, tran_a=f["t0"], tran_b=f["t1"], **kw)
        return [r.I.tolist(), r.J.tolist(), r.X.tolist()]
    ut = case.get("u_type", "INT64")
    mt = case.get("mask_type", "INT64")
    wt = case.get("w_type", et)
    if case["op"] == "mxv":
        n_out = A.ncols if f["t0"] else A.nrows
        u = O.col_vector(ut, case["u"][2], case["u"][0], _np(ut, case["u"][1]))
        w = O.col_vector(wt, n_out, *(case["w"][0], _np(wt, case["w"][1])) if "w" in case else ([], []))
        m = O.col_vector(mt, n_out, case["mask"][0], _np(mt, case["mask"][1])) if "mask" in case else None
        r = O.mxv(w, A, u, add, mul, typ, mask=m, tran_a=f["t0"], **kw)
        return [r.I.tolist(), r.X.tolist()]
    n_out = A.nrows if f["t1"] else A.ncols
    u = O.row_vector(ut, case["u"][2], case["u"][0], _np(ut, case["u"][1]))
    w = O.row_vector(wt, n_out, *(case["w"][0], _np(wt, case["w"][1])) if "w" in case else ([], []))
    m = O.row_vector(mt, n_out, case["mask"][0], _np(mt, case["mask"][1])) if "mask" in case else None
    r = O.vxm(w, u, A, add, mul, typ, mask=m, tran_a=f["t1"], **kw)
    return [r.J.tolist(), r.X.tolist()]


def run_product(case, gb):
    """The same case through pygraphblas_amd (HIP kernels), written the way the reference tests call the API."""
    T = {t.__name__: t for t in gb.types.ALL_TYPES}
    typ = case["type"]
    f = flags(case)
    et = T[case.get("expect_type", typ)]
    sr = getattr(T[typ], case["semiring"])
    desc = getattr(gb.descriptor, case["desc"]) if case.get("desc") else None
    accum = getattr(et, case["accum"]) if case.get("accum") else None
    A = gb.Matrix.from_lists(case["A"][0], case["A"][1], case["A"][2], case["A"][3], case["A"][4], typ=T[case.get("A_type", "INT64")])

    def vec(key, tkey, default_t, n):
        if key not in case:
            return None
        I, X, size = case[key]
        t = T[case.get(tkey, default_t)]
        v = gb.Vector.sparse(t, size)
        for i, x in zip(I, X):
            v[i] = x
        return v

    if case["op"] == "mxm":
        B = gb.Matrix.from_lists(case["B"][0], case["B"][1], case["B"][2], case["B"][3], case["B"][4], typ=T[case.get("B_type", "INT64")])
        out = None
        if "C" in case:
            nr = A.ncols if f["t0"] else A.nrows
            nc = B.nrows if f["t1"] else B.ncols
            out = gb.Matrix.from_lists(case["C"][0], case["C"][1], case["C"][2], nr, nc, typ=et)
        r = A.mxm(B, semiring=sr, out=out, accum=accum, desc=desc, cast=et if out is None else None)
        return [x.tolist() for x in r.to_arrays()]
    u = vec("u", "u_type", "INT64", None)
    w = vec("w", "w_type", et.__name__, None)
    m = vec("mask", "mask_type", "INT64", None)
    if case["op"] == "mxv":
        r = A.mxv(u, semiring=sr, out=w, mask=m, accum=accum, desc=desc, cast=et if w is None else None)
    else:
        if w is None:
            n_out = A.nrows if f["t1"] else A.ncols
            w = gb.Vector.sparse(et, n_out)
        r = u.vxm(A, semiring=sr, out=w, mask=m, accum=accum, desc=desc)
    return [x.tolist() for x in r.to_arrays()]
