"""Shared test helpers: random operands, conversion between the product objects and oracle tuples."""
import numpy as np

from oracle import oracle as O
import pygraphblas_amd as gb

TYPE = {t.__name__: t for t in gb.types.ALL_TYPES}
INT_TYPES = ["INT8", "UINT8", "INT16", "UINT16", "INT32", "UINT32", "INT64", "UINT64"]


def rand_values(rng, typ, n, small=True):
    """Values for type name `typ`; `small` keeps floats on a 1/8 grid so every summation order is exact."""
    if typ == "BOOL":
        return rng.integers(0, 2, n).astype(np.bool_)
    if typ in ("FP32", "FP64"):
        if small:
            return (rng.integers(-16, 17, n) / 8.0).astype(O.NP[typ])
        return rng.random(n).astype(O.NP[typ])
    info = np.iinfo(O.NP[typ])
    lo, hi = (max(info.min, -50), min(info.max, 50)) if small else (info.min, info.max)
    return rng.integers(lo, hi, n, endpoint=True).astype(O.NP[typ])


def rand_matrix(rng, typ, nrows, ncols, density, small=True):
    nnz = int(round(nrows * ncols * density))
    flat = rng.choice(nrows * ncols, size=min(nnz, nrows * ncols), replace=False) if nrows * ncols else np.zeros(0, np.int64)
    flat.sort()
    I, J = np.divmod(flat.astype(np.uint64), np.uint64(max(ncols, 1)))
    return O.Tuples(typ, nrows, ncols, I, J, rand_values(rng, typ, len(flat), small))


def rand_vector(rng, typ, n, density, small=True):
    k = int(round(n * density))
    idx = np.sort(rng.choice(n, size=min(k, n), replace=False)).astype(np.uint64) if n else np.zeros(0, np.uint64)
    return idx, rand_values(rng, typ, len(idx), small)


def to_matrix(t):
    return gb.Matrix.from_arrays(t.I, t.J, t.X, t.nrows, t.ncols, TYPE[t.typ])


def to_vector(typ, n, idx, vals):
    return gb.Vector.from_arrays(idx, vals, n, TYPE[typ])


def matrix_tuples(m):
    I, J, X = m.to_arrays()
    return O.Tuples(m.type.__name__, m.nrows, m.ncols, I, J, X)


def vector_pairs(v):
    I, X = v.to_arrays()
    return I, X


def assert_same(typ, got_idx, got_val, exp_idx, exp_val, rtol=0.0, what=""):
    assert np.array_equal(np.asarray(got_idx, np.uint64), np.asarray(exp_idx, np.uint64)), f"pattern differs {what}"
    got_val, exp_val = np.asarray(got_val), np.asarray(exp_val)
    if typ in ("FP32", "FP64") and rtol > 0:
        # the tolerance the north star states: 1e-6 relative for floating point
        assert np.allclose(got_val, exp_val, rtol=rtol, atol=0.0, equal_nan=True), f"values differ {what}"
    else:
        eq = np.array_equal(got_val, exp_val, equal_nan=True) if got_val.dtype.kind == 'f' else np.array_equal(got_val, exp_val)
        if not eq:
            bad = np.flatnonzero(~((got_val == exp_val) | ((got_val != got_val) & (exp_val != exp_val))))
            what = f"{what} first diffs at {bad[:6].tolist()}: got {got_val[bad[:6]]} expected {exp_val[bad[:6]]}"
        assert eq, f"values differ (bit-exact required) {what}: {got_val[:8]} vs {exp_val[:8]}"
