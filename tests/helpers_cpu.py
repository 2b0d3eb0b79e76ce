"""Random operand builders that need only numpy + the oracle (no HIP library) — for the `-m "not gpu"` tests."""
import numpy as np

from oracle import oracle as O


def rand_values(rng, typ, n, small=True):
    if typ == "BOOL":
        return rng.integers(0, 2, n).astype(np.bool_)
    if typ in ("FP32", "FP64"):
        return (rng.integers(-16, 17, n) / 8.0).astype(O.NP[typ]) if small else rng.random(n).astype(O.NP[typ])
    info = np.iinfo(O.NP[typ])
    lo, hi = (max(info.min, -50), min(info.max, 50)) if small else (info.min, info.max)
    return rng.integers(lo, hi, n, endpoint=True).astype(O.NP[typ])


def rand_matrix(rng, typ, nrows, ncols, density, small=True):
    nnz = int(round(nrows * ncols * density))
    flat = np.sort(rng.choice(nrows * ncols, size=min(nnz, nrows * ncols), replace=False)) if nrows * ncols else np.zeros(0, np.int64)
    I, J = np.divmod(flat.astype(np.uint64), np.uint64(max(ncols, 1)))
    return O.Tuples(typ, nrows, ncols, I, J, rand_values(rng, typ, len(flat), small))


def rand_vector(rng, typ, n, density, small=True):
    k = int(round(n * density))
    idx = np.sort(rng.choice(n, size=min(k, n), replace=False)).astype(np.uint64) if n else np.zeros(0, np.uint64)
    return idx, rand_values(rng, typ, len(idx), small)
