"""A deliberately small model of the GraphBLAS companion operations on dictionaries (test infrastructure): what the C API 1.3 says eWiseAdd / eWiseMult /
apply / select / transpose / reduce / pattern / cast do to stored entries, written independently of the HIP kernels.  It is pinned to the reference's own
test vectors (tests/golden/reference_companion_vectors.json) by tests/test_companion_golden.py — as oracle/grb_oracle.c is for the products — and the GPU
replays compare the HIP library with the same vectors directly."""
import json
import math
import os

INT_BITS = {"INT8": 8, "INT16": 16, "INT32": 32, "INT64": 64, "UINT8": 8, "UINT16": 16, "UINT32": 32, "UINT64": 64}
HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    with open(os.path.join(HERE, "golden", "reference_companion_vectors.json")) as f:
        return json.load(f)["cases"]


def wrap(typ, x):
    """A Python number as a value of the GraphBLAS type (C casts; integers wrap modulo 2^n)."""
    if typ == "BOOL":
        return bool(x)
    if typ.startswith("FP"):
        return float(x)
    b = INT_BITS[typ]; x = int(x) & ((1 << b) - 1)
    return x - (1 << b) if typ[0] == "I" and x >= 1 << (b - 1) else x


def binop(op, typ, a, b):
    if op == "PLUS": r = a + b
    elif op == "MINUS": r = a - b
    elif op == "TIMES": r = a * b
    elif op == "DIV":
        if typ.startswith("FP"): r = a / b if b else (math.inf if a > 0 else -math.inf if a < 0 else math.nan)
        else: r = int(a / b)                      # C integer division truncates toward zero (no zero divisors in the vectors)
    elif op == "FIRST": r = a
    elif op == "SECOND": r = b
    else: raise ValueError(op)
    return wrap(typ, r)


def unop(op, typ, a):
    if op == "AINV": r = -a
    elif op == "ABS": r = abs(a)
    elif op == "MINV": r = (1.0 / a if a else math.inf) if typ.startswith("FP") else (0 if a == 0 else int(1 / a))
    elif op == "IDENTITY": r = a
    elif op == "ONE": r = 1
    else: raise ValueError(op)
    return wrap(typ, r)


def keep(sel, thunk, pos, x):
    i, j = pos if isinstance(pos, tuple) else (pos, 0)
    t = thunk if thunk is not None else 0
    return {"NONZERO": x != 0, "!=0": x != 0, ">=0": x >= 0, "!=": x != t, ">": x > t, "<": x < t, ">=": x >= t, "TRIL": j <= i + t, "TRIU": j >= i + t,
            "DIAG": j == i + t, "OFFDIAG": j != i + t}[sel]


def operand(case, key):
    o = case[key]
    if case["kind"] == "matrix":
        return {(int(i), int(j)): wrap(case["type"], x) for i, j, x in zip(o[0], o[1], o[2])}
    return {int(i): wrap(case["type"], x) for i, x in zip(o[0], o[1])}


def run(case):
    """The result of one case as a sorted tuple list (or a scalar for `reduce`)."""
    typ = case["type"]; A = operand(case, "A"); op = case["op"]
    if op in ("eadd", "emult"):
        B = operand(case, "B"); out = {}
        for p in (set(A) | set(B)) if op == "eadd" else (set(A) & set(B)):
            out[p] = binop(case["binop"], typ, A[p], B[p]) if p in A and p in B else A.get(p, B.get(p))
    elif op == "apply":
        ot = case.get("unop_type", typ)          # the operator's own type: the entries are cast into it and the result back (C API 1.3 apply)
        out = {p: wrap(typ, unop(case["unop"], ot, wrap(ot, x))) for p, x in A.items()}
    elif op == "apply_first":
        out = {p: binop(case["binop"], typ, wrap(typ, case["scalar"]), x) for p, x in A.items()}
    elif op == "apply_second":
        out = {p: binop(case["binop"], typ, x, wrap(typ, case["scalar"])) for p, x in A.items()}
    elif op == "select":
        out = {p: x for p, x in A.items() if keep(case["select"], case.get("thunk"), p, x)}
    elif op == "transpose":
        out = dict(A) if "T0" in (case.get("desc") or "") else {(j, i): x for (i, j), x in A.items()}       # (the transpose of the transposed input)
    elif op == "pattern":
        out = {p: wrap(case.get("to", "BOOL"), 1) for p in A}
    elif op == "cast":
        out = {p: wrap(case["to"], x) for p, x in A.items()}
    elif op == "reduce_vector":
        out = {}
        for (i, j), x in sorted(A.items()):
            out[i] = binop(case["monoid"], typ, out[i], x) if i in out else x
    elif op == "reduce":
        ident = {"PLUS": 0, "TIMES": 1, "LOR": False, "LAND": True}[case["monoid"]]
        acc = wrap(typ, ident)
        for _, x in sorted(A.items()):
            acc = (acc or x) if case["monoid"] == "LOR" else (acc and x) if case["monoid"] == "LAND" else binop(case["monoid"], typ, acc, x)
        return wrap(case["to"], acc)
    else:
        raise ValueError(op)
    return sorted(out.items())


def expected(case):
    e = case["expect"]
    if case["op"] == "reduce":
        return e
    et = case.get("expect_type", case["type"])
    if case["kind"] == "matrix" and case["op"] != "reduce_vector":
        return sorted(((int(i), int(j)), wrap(et, x)) for i, j, x in zip(e[0], e[1], e[2]))
    return sorted((int(i), wrap(et, x)) for i, x in zip(e[0], e[1]))
