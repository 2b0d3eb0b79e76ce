"""Differential fuzzing of the host-mirror index operations (grb_host_ops.cpp) against a Python model of the GraphBLAS
semantics, driven through the UNMODIFIED reference package on top of shim/ — no GPU needed (these operations edit the host
mirror).  Run under /opt/conda/bin/python3.9 with PYTHONPATH=<repo>/shim:<reference>:

    python3.9 tools/fuzz_host_ops.py [--cases 2000] [--seed 1]

Covered: GrB_Matrix_extract, GrB_Col_extract (rows and columns), GrB_Vector_extract, GrB_Matrix_assign, GrB_Row_assign,
GrB_Col_assign, GrB_Vector_assign — index lists (with repeats for extract), ranges, ALL; valued / structural / complemented
masks; accumulators; replace; transposed input for the extracts."""
import argparse
import random

from pygraphblas import Matrix, Vector, INT64, BOOL, descriptor

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=2000); ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rnd = random.Random(args.seed)
ACC = {None: None, "PLUS": lambda a, b: a + b, "MIN": min, "SECOND": lambda a, b: b, "FIRST": lambda a, b: a}


def rand_mat(nr, nc, dens, typ=INT64, vals=lambda: rnd.randint(-9, 9)):
    M = Matrix.sparse(typ, nr, nc); d = {}
    for i in range(nr):
        for j in range(nc):
            if rnd.random() < dens:
                x = vals(); M[i, j] = x; d[(i, j)] = x
    return M, d


def rand_vec(n, dens, typ=INT64, vals=lambda: rnd.randint(-9, 9)):
    v = Vector.sparse(typ, n); d = {}
    for i in range(n):
        if rnd.random() < dens:
            x = vals(); v[i] = x; d[i] = x
    return v, d


def rand_index(n, repeats):
    """(argument for the reference, the list of source indices it denotes)"""
    k = rnd.random()
    if k < 0.3:
        return None, list(range(n))
    if k < 0.6 and n > 1:
        a, b = sorted(rnd.sample(range(n), 2)); return slice(a, b), list(range(a, b + 1))
    size = rnd.randint(1, n)
    lst = [rnd.randrange(n) for _ in range(size)] if repeats else rnd.sample(range(n), size)
    return lst, lst


def desc_of(replace, struct, comp, t0=False):
    d = None                       # composed with `&` as the reference does itself (not every combination has a module-level name)
    for on, flag in ((replace, descriptor.R), (struct, descriptor.S), (comp, descriptor.C), (t0, descriptor.T0)):
        if on:
            d = flag if d is None else d & flag
    return d


def allows(mask, p, struct, comp):
    if mask is None:
        return not comp
    t = p in mask and (struct or bool(mask[p]))
    return t != comp


def finish(C, Z, space, mask, struct, comp, replace, scope=lambda p: True):
    out = {}
    for p in space:
        if not scope(p):
            if p in C: out[p] = C[p]
        elif allows(mask, p, struct, comp):
            if p in Z: out[p] = Z[p]
        elif not replace and p in C:
            out[p] = C[p]
    return out


def accumulate(C, T, acc):
    if acc is None:
        return dict(T)
    Z = dict(C)
    for p, x in T.items():
        Z[p] = ACC[acc](Z[p], x) if p in Z else x
    return Z


def mdict(M): return {(i, j): x for i, j, x in M}
def vdict(v): return {i: x for i, x in v}


def maskargs(shape_rand):
    """a mask (or none) with its flags"""
    if rnd.random() < 0.4:
        return None, None, False, False
    M, d = shape_rand()
    return M, d, rnd.random() < 0.4, rnd.random() < 0.4


n_done = 0
for case in range(args.cases):
    nr, nc = rnd.randint(1, 7), rnd.randint(1, 7)
    acc = rnd.choice(list(ACC)); accop = getattr(INT64, acc) if acc else None
    replace = rnd.random() < 0.3
    kind = rnd.choice(["mextract", "cextract", "vextract", "massign", "rassign", "cassign", "vassign", "hvscalar", "hmscalar"])
    bvals = lambda: rnd.random() < 0.7
    if kind == "mextract":
        t0 = rnd.random() < 0.3
        A, a = rand_mat(nr, nc, 0.5)
        sr, sc = (nc, nr) if t0 else (nr, nc)                         # shape of op(A)
        Iarg, I = rand_index(sr, True); Jarg, J = rand_index(sc, True)
        C, c = rand_mat(len(I), len(J), 0.4)
        M, m, struct, comp = maskargs(lambda: rand_mat(len(I), len(J), 0.5, BOOL, bvals))
        A.extract_matrix(Iarg, Jarg, out=C, mask=M, accum=accop, desc=desc_of(replace, struct, comp, t0))
        src = {(j, i): x for (i, j), x in a.items()} if t0 else a
        T = {(k, l): src[(I[k], J[l])] for k in range(len(I)) for l in range(len(J)) if (I[k], J[l]) in src}
        space = [(k, l) for k in range(len(I)) for l in range(len(J))]
        exp = finish(c, accumulate(c, T, acc), space, m, struct, comp, replace)
        assert mdict(C) == exp, (case, kind, Iarg, Jarg, acc, replace, struct, comp, t0, mdict(C), exp)
    elif kind == "cextract":
        row = rnd.random() < 0.5                                      # extract_row is GrB_Col_extract of the transpose
        A, a = rand_mat(nr, nc, 0.5)
        length, fixed_n = (nc, nr) if row else (nr, nc)
        f = rnd.randrange(fixed_n)
        Iarg, I = rand_index(length, True)
        if Iarg is None or isinstance(Iarg, slice):                    # the reference sizes its output by the full length for these
            I = list(range(length)) if Iarg is None else I
        w, wd = rand_vec(len(I), 0.4)
        M, m, struct, comp = maskargs(lambda: rand_vec(len(I), 0.5, BOOL, bvals))
        if row:       # the reference's extract_row passes neither mask nor accumulator on (pygraphblas/matrix.py:2962-2965) and sets T0 itself
            M, m, struct, comp, acc, accop, replace = None, None, False, False, None, None, False
        try:
            (A.extract_row if row else A.extract_col)(f, Iarg, out=w, mask=M, accum=accop, desc=desc_of(replace, struct, comp))
        except Exception as e:                                         # noqa: BLE001 - shapes the reference's wrapper itself refuses
            if "Dimension" in type(e).__name__: continue
            raise
        line = {j: x for (i, j), x in a.items() if i == f} if row else {i: x for (i, j), x in a.items() if j == f}
        T = {k: line[I[k]] for k in range(len(I)) if I[k] in line}
        exp = finish(wd, accumulate(wd, T, acc), range(len(I)), m, struct, comp, replace)
        assert vdict(w) == exp, (case, kind, row, f, Iarg, acc, replace, struct, comp, vdict(w), exp)
    elif kind == "vextract":
        u, ud = rand_vec(nr, 0.5)
        Iarg, I = rand_index(nr, True)
        if Iarg is None: continue
        got = u.extract(Iarg if not isinstance(Iarg, slice) else Iarg)
        T = {k: ud[I[k]] for k in range(len(I)) if I[k] in ud}
        assert vdict(got) == T, (case, kind, Iarg, vdict(got), T)
    elif kind == "massign":
        C, c = rand_mat(nr, nc, 0.5)
        Iarg, I = rand_index(nr, False); Jarg, J = rand_index(nc, False)
        A, a = rand_mat(len(I), len(J), 0.5)
        M, m, struct, comp = maskargs(lambda: rand_mat(nr, nc, 0.5, BOOL, bvals))
        C.assign_matrix(A, Iarg, Jarg, mask=M, accum=accop, desc=desc_of(replace, struct, comp))
        Z = dict(c)
        for k in range(len(I)):
            for l in range(len(J)):
                p = (I[k], J[l])
                if (k, l) in a:
                    Z[p] = ACC[acc](Z[p], a[(k, l)]) if (acc and p in Z) else a[(k, l)]
                elif acc is None:
                    Z.pop(p, None)
        space = [(i, j) for i in range(nr) for j in range(nc)]
        exp = finish(c, Z, space, m, struct, comp, replace)
        assert mdict(C) == exp, (case, kind, Iarg, Jarg, acc, replace, struct, comp, mdict(C), exp)
    elif kind in ("rassign", "cassign"):
        row = kind == "rassign"
        C, c = rand_mat(nr, nc, 0.5)
        length, fixed_n = (nc, nr) if row else (nr, nc)
        f = rnd.randrange(fixed_n)
        Jarg, J = rand_index(length, False)
        u, ud = rand_vec(len(J), 0.5)
        M, m, struct, comp = maskargs(lambda: rand_vec(length, 0.5, BOOL, bvals))
        (C.assign_row if row else C.assign_col)(f, u, Jarg, mask=M, accum=accop, desc=desc_of(replace, struct, comp))
        pos = (lambda k: (f, k)) if row else (lambda k: (k, f))
        Z = dict(c)
        for k in range(len(J)):
            p = pos(J[k])
            if k in ud:
                Z[p] = ACC[acc](Z[p], ud[k]) if (acc and p in Z) else ud[k]
            elif acc is None:
                Z.pop(p, None)
        space = [(i, j) for i in range(nr) for j in range(nc)]
        mm = None if m is None else {pos(k): x for k, x in m.items()}
        scope = (lambda p: p[0] == f) if row else (lambda p: p[1] == f)
        exp = finish(c, Z, space, mm, struct, comp, replace, scope)
        assert mdict(C) == exp, (case, kind, f, Jarg, acc, replace, struct, comp, mdict(C), exp)
    elif kind in ("hvscalar", "hmscalar"):
        # a scalar assigned into a hypersparse container (2^60 dimensions: host-side bookkeeping, grb_host_ops.cpp host_assign_scalar):
        # either everywhere under a non-complemented mask (the pattern is bounded by the mask's), or over a short index list
        IMAX = 1 << 60
        pool = sorted({rnd.randrange(IMAX) for _ in range(6)} | {0, IMAX - 1})
        s_val = rnd.randint(-5, 5)
        if kind == "hvscalar":
            w = Vector.sparse(INT64, IMAX); wd = {}
            for p in pool:
                if rnd.random() < 0.5: x = rnd.randint(-9, 9); w[p] = x; wd[p] = x
            use_list = rnd.random() < 0.5
            M = Vector.sparse(BOOL, IMAX); m = {}
            for p in pool:
                if rnd.random() < 0.6: x = rnd.random() < 0.7; M[p] = x; m[p] = x
            struct = rnd.random() < 0.4; comp = use_list and rnd.random() < 0.4
            idx = rnd.sample(pool, rnd.randint(1, len(pool))) if use_list else None
            w.assign_scalar(s_val, idx, mask=M, accum=accop, desc=desc_of(replace, struct, comp))
            region = idx if use_list else [p for p in pool]              # (ALL: only positions the mask allows can change, all of them in the pool)
            Z = dict(wd)
            for p in region:
                Z[p] = ACC[acc](Z[p], s_val) if (acc and p in Z) else s_val
            exp = finish(wd, Z, pool, m, struct, comp, replace)
            assert vdict(w) == exp, (case, kind, idx, acc, replace, struct, comp, wd, m, vdict(w), exp)
        else:
            C = Matrix.sparse(INT64, IMAX, IMAX); c = {}; M = Matrix.sparse(BOOL, IMAX, IMAX); m = {}
            cells = [(a, b) for a in pool[:4] for b in pool[-4:]]
            for p in cells:
                if rnd.random() < 0.4: x = rnd.randint(-9, 9); C[p] = x; c[p] = x
                if rnd.random() < 0.5: x = rnd.random() < 0.7; M[p] = x; m[p] = x
            struct = rnd.random() < 0.4
            C.assign_scalar(s_val, mask=M, accum=accop, desc=desc_of(replace, struct, False))
            Z = dict(c)
            for p in cells:
                Z[p] = ACC[acc](Z[p], s_val) if (acc and p in Z) else s_val
            exp = finish(c, Z, cells, m, struct, False, replace)
            assert mdict(C) == exp, (case, kind, acc, replace, struct, c, m, mdict(C), exp)
    else:   # vassign
        w, wd = rand_vec(nr, 0.5)
        Iarg, I = rand_index(nr, False)
        u, ud = rand_vec(len(I), 0.5)
        M, m, struct, comp = maskargs(lambda: rand_vec(nr, 0.5, BOOL, bvals))
        w.assign(u, Iarg, mask=M, accum=accop, desc=desc_of(replace, struct, comp))
        Z = dict(wd)
        for k in range(len(I)):
            if k in ud:
                Z[I[k]] = ACC[acc](Z[I[k]], ud[k]) if (acc and I[k] in Z) else ud[k]
            elif acc is None:
                Z.pop(I[k], None)
        exp = finish(wd, Z, range(nr), m, struct, comp, replace)
        assert vdict(w) == exp, (case, kind, Iarg, acc, replace, struct, comp, vdict(w), exp)
    n_done += 1
print(f"fuzz host ops ok: {n_done} cases, seed {args.seed}")
