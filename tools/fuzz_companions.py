"""Differential fuzzing of the device companions of the hot path (eWiseAdd / eWiseMult, apply, select, transpose,
reduce to a vector, scalar assign — vectors and matrices) against a Python model of the GraphBLAS rules, through the
ctypes mirror.  Needs the GPU (these are HIP kernels):

    python tools/fuzz_companions.py [--seconds 60] [--seed 1]

Every case draws shapes, operands, a mask (valued / structural / complemented / none), an accumulator, replace and, for the
matrix operations, transposed inputs; integer values, so every comparison is exact."""
import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pygraphblas_amd as gb  # noqa: E402
from pygraphblas_amd import descriptor as D  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=60); ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rnd = random.Random(args.seed)
T = gb.INT64
BIN = {"PLUS": lambda a, b: a + b, "MIN": min, "MAX": max, "TIMES": lambda a, b: a * b, "FIRST": lambda a, b: a, "SECOND": lambda a, b: b, "MINUS": lambda a, b: a - b}
UN = {"AINV": lambda a: -a, "ABS": abs, "IDENTITY": lambda a: a, "ONE": lambda a: 1}


def rand_mat(nr, nc, dens, typ=T, vals=lambda: rnd.randint(-9, 9)):
    d = {(i, j): vals() for i in range(nr) for j in range(nc) if rnd.random() < dens}
    I = [p[0] for p in d]; J = [p[1] for p in d]; V = list(d.values())
    M = gb.Matrix.from_lists(I, J, V, nr, nc, typ) if d else gb.Matrix.sparse(typ, nr, nc)
    return M, d


def rand_vec(n, dens, typ=T, vals=lambda: rnd.randint(-9, 9)):
    d = {i: vals() for i in range(n) if rnd.random() < dens}
    v = gb.Vector.from_lists(list(d), list(d.values()), n, typ) if d else gb.Vector.sparse(typ, n)
    return v, d


def desc_of(replace, struct, comp, t0=False, t1=False):
    name = ("R" if replace else "") + ("S" if struct else "") + ("C" if comp else "") + ("T0" if t0 else "") + ("T1" if t1 else "")
    return getattr(D, name) if name else None


def allows(mask, p, struct, comp):
    if mask is None:
        return not comp
    return (p in mask and (struct or bool(mask[p]))) != comp


def finish(C, Tn, space, mask, struct, comp, replace, acc):
    Z = dict(Tn) if acc is None else dict(C)
    if acc is not None:
        for p, x in Tn.items():
            Z[p] = BIN[acc](Z[p], x) if p in Z else x
    out = {}
    for p in space:
        if allows(mask, p, struct, comp):
            if p in Z: out[p] = Z[p]
        elif not replace and p in C:
            out[p] = C[p]
    return out


def mdict(M):
    I, J, X = M.to_arrays(); return {(int(i), int(j)): int(x) for i, j, x in zip(I, J, X)}


def vdict(v):
    I, X = v.to_arrays(); return {int(i): int(x) for i, x in zip(I, X)}


def tr(d): return {(j, i): x for (i, j), x in d.items()}


t_end = time.time() + args.seconds
n = 0
counts = {}
while time.time() < t_end:
    n += 1
    nr, nc = rnd.randint(1, 9), rnd.randint(1, 9)
    acc = rnd.choice([None, None, "PLUS", "MIN", "SECOND"]); accop = getattr(T, acc) if acc else None
    replace = rnd.random() < 0.3
    use_mask = rnd.random() < 0.6
    struct, comp = (rnd.random() < 0.4, rnd.random() < 0.4) if use_mask else (False, False)
    bvals = lambda: rnd.random() < 0.7
    kind = rnd.choice(["vewise", "mewise", "vapply", "mapply", "select", "transpose", "reducev", "vassign", "massign", "vbind"])
    counts[kind] = counts.get(kind, 0) + 1
    what = (n, kind, acc, replace, use_mask, struct, comp)
    if kind in ("vewise", "vapply", "vassign", "vbind"):
        w, wd = rand_vec(nr, 0.4); M, m = rand_vec(nr, 0.5, gb.BOOL, bvals) if use_mask else (None, None)
        d = desc_of(replace, struct, comp); space = range(nr)
        if kind == "vewise":
            u, ud = rand_vec(nr, 0.5); v, vd = rand_vec(nr, 0.5); op = rnd.choice(list(BIN)); union = rnd.random() < 0.5
            (u.eadd if union else u.emult)(v, getattr(T, op), out=w, mask=M, accum=accop, desc=d)
            Tn = {p: (BIN[op](ud[p], vd[p]) if p in ud and p in vd else (ud[p] if p in ud else vd[p])) for p in (set(ud) | set(vd) if union else set(ud) & set(vd))}
            what += (op, union)
        elif kind == "vapply":
            u, ud = rand_vec(nr, 0.5); op = rnd.choice(list(UN))
            u.apply(getattr(T, op), out=w, mask=M, accum=accop, desc=d); Tn = {p: UN[op](x) for p, x in ud.items()}; what += (op,)
        elif kind == "vbind":
            u, ud = rand_vec(nr, 0.5); op = rnd.choice(["PLUS", "MINUS", "TIMES", "MIN"]); s = rnd.randint(-5, 5); first = rnd.random() < 0.5
            (u.apply_first(s, getattr(T, op), out=w, mask=M, accum=accop, desc=d) if first else u.apply_second(getattr(T, op), s, out=w, mask=M, accum=accop, desc=d))
            Tn = {p: (BIN[op](s, x) if first else BIN[op](x, s)) for p, x in ud.items()}; what += (op, s, first)
        else:
            s = rnd.randint(-5, 5); idx = None if rnd.random() < 0.5 else rnd.sample(range(nr), rnd.randint(1, nr))
            w.assign_scalar(s, idx, mask=M, accum=accop, desc=d)
            Z = dict(wd)
            for p in (range(nr) if idx is None else idx):
                Z[p] = BIN[acc](Z[p], s) if (acc and p in Z) else s
            exp = {}
            for p in space:
                if allows(m, p, struct, comp):
                    if p in Z: exp[p] = Z[p]
                elif not replace and p in wd:
                    exp[p] = wd[p]
            assert vdict(w) == exp, (what, idx, s, wd, m, vdict(w), exp); continue
        exp = finish(wd, Tn, space, m, struct, comp, replace, acc)
        assert vdict(w) == exp, (what, wd, m, vdict(w), exp, gb.last_kernel_plan())
    else:
        t0 = rnd.random() < 0.3; t1 = rnd.random() < 0.3
        C, c = rand_mat(nr, nc, 0.4); M, m = rand_mat(nr, nc, 0.5, gb.BOOL, bvals) if use_mask else (None, None)
        space = [(i, j) for i in range(nr) for j in range(nc)]
        if kind == "mewise":
            A, a = rand_mat(*((nc, nr) if t0 else (nr, nc)), 0.5); B, b = rand_mat(*((nc, nr) if t1 else (nr, nc)), 0.5)
            op = rnd.choice(list(BIN)); union = rnd.random() < 0.5
            (A.eadd if union else A.emult)(B, getattr(T, op), out=C, mask=M, accum=accop, desc=desc_of(replace, struct, comp, t0, t1))
            a2, b2 = (tr(a) if t0 else a), (tr(b) if t1 else b)
            Tn = {p: (BIN[op](a2[p], b2[p]) if p in a2 and p in b2 else (a2[p] if p in a2 else b2[p])) for p in (set(a2) | set(b2) if union else set(a2) & set(b2))}
            what += (op, union, t0, t1)
        elif kind == "mapply":
            A, a = rand_mat(*((nc, nr) if t0 else (nr, nc)), 0.5); op = rnd.choice(list(UN))
            A.apply(getattr(T, op), out=C, mask=M, accum=accop, desc=desc_of(replace, struct, comp, t0))
            Tn = {p: UN[op](x) for p, x in (tr(a) if t0 else a).items()}; what += (op, t0)
        elif kind == "select":
            A, a = rand_mat(*((nc, nr) if t0 else (nr, nc)), 0.6); sel = rnd.choice(["TRIL", "TRIU", "DIAG", "OFFDIAG", "NONZERO", "GT_THUNK", "LE_THUNK", "EQ_THUNK", "GT_ZERO"])
            k = rnd.randint(-3, 3)
            A.select(sel, None if sel in ("NONZERO", "GT_ZERO") else k, out=C, mask=M, accum=accop, desc=desc_of(replace, struct, comp, t0))
            keep = {"TRIL": lambda i, j, x: j - i <= k, "TRIU": lambda i, j, x: j - i >= k, "DIAG": lambda i, j, x: j - i == k, "OFFDIAG": lambda i, j, x: j - i != k,
                    "NONZERO": lambda i, j, x: x != 0, "GT_THUNK": lambda i, j, x: x > k, "LE_THUNK": lambda i, j, x: x <= k, "EQ_THUNK": lambda i, j, x: x == k, "GT_ZERO": lambda i, j, x: x > 0}[sel]
            Tn = {p: x for p, x in (tr(a) if t0 else a).items() if keep(p[0], p[1], x)}; what += (sel, k, t0)
        elif kind == "transpose":
            A, a = rand_mat(*((nr, nc) if t0 else (nc, nr)), 0.5)
            A.transpose(out=C, mask=M, accum=accop, desc=desc_of(replace, struct, comp, t0))
            Tn = a if t0 else tr(a); what += (t0,)                     # (T0 on transpose: the transpose of the transpose)
        elif kind == "reducev":
            A, a = rand_mat(*((nc, nr) if t0 else (nr, nc)), 0.5); mon = rnd.choice(["PLUS", "MIN", "MAX"])
            w, wd = rand_vec(nr, 0.4); Mv, mv = rand_vec(nr, 0.5, gb.BOOL, bvals) if use_mask else (None, None)
            A.reduce_vector(getattr(T, mon + "_MONOID"), out=w, mask=Mv, accum=accop, desc=desc_of(replace, struct, comp, t0))
            Tn = {}
            for (i, j), x in (tr(a) if t0 else a).items():
                Tn[i] = BIN[mon](Tn[i], x) if i in Tn else x
            exp = finish(wd, Tn, range(nr), mv, struct, comp, replace, acc)
            assert vdict(w) == exp, (what, mon, t0, a, wd, mv, vdict(w), exp); continue
        else:   # massign: a scalar into a region of C
            s = rnd.randint(-5, 5)
            I = None if rnd.random() < 0.4 else sorted(rnd.sample(range(nr), rnd.randint(1, nr))); J = None if rnd.random() < 0.4 else sorted(rnd.sample(range(nc), rnd.randint(1, nc)))
            fn = getattr(gb.lib, "GrB_Matrix_assign_INT64")
            import ctypes as Ct
            from pygraphblas_amd.base import check
            from pygraphblas_amd.matrix import get_args
            mh, ah, dh = get_args(M, accop, desc_of(replace, struct, comp))
            ALL = Ct.cast(gb._capi.handle("GrB_ALL"), Ct.c_void_p)
            Ia = np.ascontiguousarray(I if I is not None else [], np.uint64); Ja = np.ascontiguousarray(J if J is not None else [], np.uint64)
            check(fn(C._h, mh, ah, Ct.c_int64(s), ALL if I is None else Ia.ctypes.data_as(Ct.c_void_p), Ct.c_uint64(0 if I is None else len(I)),
                     ALL if J is None else Ja.ctypes.data_as(Ct.c_void_p), Ct.c_uint64(0 if J is None else len(J)), dh), C)
            Z = dict(c)
            for i in (range(nr) if I is None else I):
                for j in (range(nc) if J is None else J):
                    Z[(i, j)] = BIN[acc](Z[(i, j)], s) if (acc and (i, j) in Z) else s
            exp = {}
            for p in space:
                if allows(m, p, struct, comp):
                    if p in Z: exp[p] = Z[p]
                elif not replace and p in c:
                    exp[p] = c[p]
            assert mdict(C) == exp, (what, I, J, s, c, m, mdict(C), exp); continue
        exp = finish(c, Tn, space, m, struct, comp, replace, acc)
        assert mdict(C) == exp, (what, c, m, mdict(C), exp, gb.last_kernel_plan())
print(f"fuzz companions ok: {n} cases in {args.seconds:.0f} s, seed {args.seed}: {counts}")
