"""Synthetic R-MAT / Graph500 Kronecker graphs for the BASELINE.json configs.

scale S, edgefactor 16 => m = 16 * 2^S sampled edges on n = 2^S vertices, quadrant
probabilities (a, b, c, d) = (0.57, 0.19, 0.19, 0.05).  The generator is *counter based*: edge e,
bit level l draw a 16-bit uniform from mix64(seed, e, l // 4), so any slice of the edge list can be
produced independently (one rank of a row-partitioned run generates only what it needs to look at),
on the CPU with numpy or in HBM with torch, bit-identically (tests/test_rmat.py).
Duplicates collapse to one entry; `symmetric=True` gives A ∪ Aᵀ; `drop_self_loops` as the config asks.
This is workload synthesis for bench.py / the tests — not part of the GraphBLAS API surface.
"""
import numpy as np

_A16, _AB16, _ABC16 = 37356, 49807, 62259      # 0.57, 0.76, 0.95 of 2^16
_K1, _K2, _K3 = 0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB
_MASK = (1 << 64) - 1


def _mix64_np(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(_K2)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(_K3)
    return z ^ (z >> np.uint64(31))


def edges_numpy(scale, seed=42, edgefactor=16, first=0, count=None):
    """(src, dst) uint64 arrays for edges [first, first+count) of the scale-`scale` R-MAT stream."""
    m = edgefactor << scale
    if count is None:
        count = m - first
    with np.errstate(over="ignore"):
        e = np.arange(first, first + count, dtype=np.uint64)
        base = e * np.uint64(_K1) + np.uint64((seed * 0xD6E8FEB86659FD93) & _MASK)
        src = np.zeros(count, np.uint64)
        dst = np.zeros(count, np.uint64)
        h = None
        for lvl in range(scale):
            if lvl % 4 == 0:
                h = _mix64_np(base + np.uint64(((lvl // 4 + 1) * _K3) & _MASK))
            r = (h >> np.uint64(16 * (lvl % 4))) & np.uint64(0xFFFF)
            ibit = (r >= _AB16).astype(np.uint64)
            jbit = (((r >= _A16) & (r < _AB16)) | (r >= _ABC16)).astype(np.uint64)
            src |= ibit << np.uint64(lvl)
            dst |= jbit << np.uint64(lvl)
    return src, dst


def _lsr(z, k):   # logical shift right on int64 tensors
    import torch
    return (z >> k) & ((1 << (64 - k)) - 1)


def _s64(x):      # python int (mod 2^64) -> signed int64 value
    x &= _MASK
    return x - (1 << 64) if x >= (1 << 63) else x


def edges_torch(scale, device, seed=42, edgefactor=16, first=0, count=None):
    """Same stream as edges_numpy, computed on `device` with int64 two's-complement arithmetic."""
    import torch
    m = edgefactor << scale
    if count is None:
        count = m - first
    e = torch.arange(first, first + count, dtype=torch.int64, device=device)
    base = e * _s64(_K1) + _s64(seed * 0xD6E8FEB86659FD93)
    src = torch.zeros(count, dtype=torch.int64, device=device)
    dst = torch.zeros(count, dtype=torch.int64, device=device)
    h = None
    for lvl in range(scale):
        if lvl % 4 == 0:
            z = base + _s64((lvl // 4 + 1) * _K3)
            z = (z ^ _lsr(z, 30)) * _s64(_K2)
            z = (z ^ _lsr(z, 27)) * _s64(_K3)
            h = z ^ _lsr(z, 31)
        r = _lsr(h, 16 * (lvl % 4)) & 0xFFFF if lvl % 4 else h & 0xFFFF
        ibit = (r >= _AB16).to(torch.int64)
        jbit = (((r >= _A16) & (r < _AB16)) | (r >= _ABC16)).to(torch.int64)
        src |= ibit << lvl
        dst |= jbit << lvl
    return src, dst


def _finish_numpy(src, dst, n, symmetric, drop_self_loops, lower, row_range):
    if symmetric:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    keep = np.ones(len(src), bool)
    if drop_self_loops:
        keep &= src != dst
    if lower:
        keep &= dst < src
    if row_range is not None:
        keep &= (src >= row_range[0]) & (src < row_range[1])
    src, dst = src[keep], dst[keep]
    key = np.unique((src << np.uint64(32)) | dst)
    rows = (key >> np.uint64(32)).astype(np.int64)
    cols = (key & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    r0, r1 = (0, n) if row_range is None else row_range
    rowptr = np.zeros(r1 - r0 + 1, np.int64)
    np.add.at(rowptr, rows - r0 + 1, 1)
    return np.cumsum(rowptr).astype(np.uint32), cols


def permutation_numpy(scale, seed):
    """A pseudo-random relabelling of the 2^scale vertices (Graph500 permutes vertex labels; BASELINE's recipe does not):
    vertex i gets the rank of hash(seed, i).  Same permutation as permutation_torch."""
    with np.errstate(over="ignore"):
        i = np.arange(1 << scale, dtype=np.uint64)
        h = _mix64_np(i * np.uint64(_K1) + np.uint64((seed * _K3) & _MASK))
    order = np.argsort(h, kind="stable")
    perm = np.empty(1 << scale, np.uint64)
    perm[order] = np.arange(1 << scale, dtype=np.uint64)
    return perm


def permutation_torch(scale, device, seed):
    import torch
    i = torch.arange(1 << scale, dtype=torch.int64, device=device)
    z = i * _s64(_K1) + _s64(seed * _K3)
    z = (z ^ _lsr(z, 30)) * _s64(_K2)
    z = (z ^ _lsr(z, 27)) * _s64(_K3)
    h = z ^ _lsr(z, 31)
    # unsigned order of the 64-bit hash: flip the sign bit
    order = torch.argsort(h ^ _s64(1 << 63), stable=True)
    perm = torch.empty(1 << scale, dtype=torch.int64, device=device)
    perm[order] = torch.arange(1 << scale, dtype=torch.int64, device=device)
    return perm


def csr_numpy(scale, seed=42, edgefactor=16, symmetric=False, drop_self_loops=False, lower=False, row_range=None, permute_seed=None,
              transpose=False):
    """Pattern CSR (rowptr u32, col u32) of the de-duplicated R-MAT graph (rows in row_range if given).
    permute_seed: relabel the vertices pseudo-randomly first (None = keep the generator's labels).
    transpose: the CSR of A' instead of A (row_range then selects rows of A')."""
    src, dst = edges_numpy(scale, seed, edgefactor)
    if permute_seed is not None:
        perm = permutation_numpy(scale, permute_seed)
        src, dst = perm[src], perm[dst]
    if transpose:
        src, dst = dst, src
    return _finish_numpy(src, dst, 1 << scale, symmetric, drop_self_loops, lower, row_range)


def csr_torch(scale, device, seed=42, edgefactor=16, symmetric=False, drop_self_loops=False, lower=False, row_range=None,
              chunk=1 << 26, permute_seed=None, transpose=False):
    """Same CSR as csr_numpy, built in HBM: returns (rowptr int32-as-uint32 tensor, col tensor) on `device`."""
    import torch
    n = 1 << scale
    m = edgefactor << scale
    keys = []
    perm = permutation_torch(scale, device, permute_seed) if permute_seed is not None else None
    for first in range(0, m, chunk):
        s, d = edges_torch(scale, device, seed, edgefactor, first, min(chunk, m - first))
        if perm is not None:
            s, d = perm[s], perm[d]
        if transpose:
            s, d = d, s
        if symmetric:
            s, d = torch.cat([s, d]), torch.cat([d, s])
        keep = torch.ones_like(s, dtype=torch.bool)
        if drop_self_loops:
            keep &= s != d
        if lower:
            keep &= d < s
        if row_range is not None:
            keep &= (s >= row_range[0]) & (s < row_range[1])
        keys.append(((s[keep] << 32) | d[keep]))
        del s, d, keep
    key = torch.unique(torch.cat(keys), sorted=True)
    del keys
    rows = key >> 32
    cols = (key & 0xFFFFFFFF).to(torch.int32)      # bit pattern of the u32 column index
    r0, r1 = (0, n) if row_range is None else row_range
    counts = torch.bincount(rows - r0, minlength=r1 - r0)
    rowptr = torch.zeros(r1 - r0 + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32), cols


def values_numpy(nnz, seed=43, dtype=np.float64):
    """Entry values in [0, 1): counter-based as well (entry k of the sorted CSR gets hash(seed, k))."""
    with np.errstate(over="ignore"):
        k = np.arange(nnz, dtype=np.uint64)
        h = _mix64_np(k * np.uint64(_K1) + np.uint64((seed * _K2) & _MASK))
    return ((h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))).astype(dtype)


def values_torch(nnz, device, seed=43, dtype=None):
    import torch
    k = torch.arange(nnz, dtype=torch.int64, device=device)
    z = k * _s64(_K1) + _s64(seed * _K2)
    z = (z ^ _lsr(z, 30)) * _s64(_K2)
    z = (z ^ _lsr(z, 27)) * _s64(_K3)
    h = z ^ _lsr(z, 31)
    v = _lsr(h, 11).to(torch.float64) * (1.0 / (1 << 53))
    return v if dtype is None else v.to(dtype)


def csr_numpy_pcg64(scale, seed=42, edgefactor=16, symmetric=False, drop_self_loops=False, lower=False):
    """The graph of SURVEY.md §8d, to the letter: `numpy.random.Generator(PCG64(seed))`, one `random(m)` draw of doubles per bit level
    (levels outer, edges vectorised inner), u in [0, 1): i_bit = u >= a + b;  j_bit = (a <= u < a + b) or (u >= a + b + c);  bit level l sets bit l;
    no relabelling; duplicates collapse to one entry.  Same distribution as the counter-based stream above, a different graph (that one can be
    generated slice by slice in HBM, which the row-partitioned runs need; this one is a sequential stream) — kept so that the headline can be shown
    not to depend on the generator (tests/test_baseline_configs_gpu.py).  Returns (rowptr uint32[n + 1], col uint32[nnz])."""
    n = 1 << scale; m = edgefactor << scale
    a, b, c = 0.57, 0.19, 0.19
    rng = np.random.Generator(np.random.PCG64(seed))
    src = np.zeros(m, np.uint64); dst = np.zeros(m, np.uint64)
    for lvl in range(scale):
        u = rng.random(m)
        src |= (u >= a + b).astype(np.uint64) << np.uint64(lvl)
        dst |= (((u >= a) & (u < a + b)) | (u >= a + b + c)).astype(np.uint64) << np.uint64(lvl)
        del u
    return _finish_numpy(src, dst, n, symmetric, drop_self_loops, lower, None)


def values_numpy_pcg64(nnz, seed=43):
    """`Generator(PCG64(43)).random(nnz)` for the matrix values, seed 44 for the operand."""
    return np.random.Generator(np.random.PCG64(seed)).random(nnz)
