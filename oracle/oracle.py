"""ctypes front-end of oracle/liboracle.so (grb_oracle.c) — TEST INFRASTRUCTURE ONLY.

Values travel as numpy tuple arrays (I, J, X); vectors are n x 1 (mxv) or 1 x n (vxm) matrices, so
the single restated operation `C<M,r> = accum(C, op(A) add.mul op(B))` covers GrB_mxm, GrB_mxv
and GrB_vxm exactly as SURVEY.md Appendix A defines them.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

TYPES = ["BOOL", "INT8", "UINT8", "INT16", "UINT16", "INT32", "UINT32", "INT64", "UINT64", "FP32", "FP64"]
NP = {"BOOL": np.bool_, "INT8": np.int8, "UINT8": np.uint8, "INT16": np.int16, "UINT16": np.uint16, "INT32": np.int32,
      "UINT32": np.uint32, "INT64": np.int64, "UINT64": np.uint64, "FP32": np.float32, "FP64": np.float64}
OPS = ["FIRST", "SECOND", "PAIR", "ANY", "MIN", "MAX", "PLUS", "MINUS", "RMINUS", "TIMES", "DIV", "RDIV", "POW", "ISEQ", "ISNE",
       "ISGT", "ISLT", "ISGE", "ISLE", "LOR", "LAND", "LXOR", "EQ", "NE", "GT", "LT", "GE", "LE", "LXNOR"]
F_REPLACE, F_MASK_COMP, F_MASK_STRUCT, F_TRAN_A, F_TRAN_B = 1, 2, 4, 8, 16


def build():
    """Compile liboracle.so with gcc (idempotent)."""
    src = os.path.join(_HERE, "grb_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.fast_tricount_LL_maskL.restype = C.c_int64
        _lib.fast_bfs_levels.restype = C.c_int
        _lib.oracle_num_threads.restype = C.c_int
        # a container's CPU quota (cgroup v2 cpu.max) can be far below the number of visible CPUs: more OpenMP threads than
        # the quota pays for get throttled (measured on the GPU box: 256 CPUs visible, quota 16 -> 128 threads run the
        # scale-22 SpMV at 2 GFLOP/s)
        if "OMP_NUM_THREADS" not in os.environ:
            q = cpu_quota()
            if q and q < _lib.oracle_num_threads():
                _lib.oracle_set_num_threads(C.c_int(q))
    return _lib


def cpu_quota():
    """CPUs' worth of time the cgroup may use (None if unlimited / unknown)."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else max(1, int(int(quota) / int(period)))
    except Exception:
        return None


def tcode(t):
    return TYPES.index(t)


def opcode(o):
    return -1 if o is None else OPS.index(o)


def _arr(a, dt):
    return np.ascontiguousarray(np.asarray(a, dtype=dt))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Tuples:
    """A sparse matrix as sorted-or-not tuples.  Vectors: ncols == 1 (column) or nrows == 1 (row)."""

    def __init__(self, typ, nrows, ncols, I=(), J=(), X=()):
        self.typ, self.nrows, self.ncols = typ, int(nrows), int(ncols)
        self.I, self.J = _arr(I, np.uint64), _arr(J, np.uint64)
        self.X = _arr(X, NP[typ])
        assert len(self.I) == len(self.J) == len(self.X)

    @property
    def nvals(self):
        return len(self.I)

    def sorted(self):
        o = np.lexsort((self.J, self.I))
        return Tuples(self.typ, self.nrows, self.ncols, self.I[o], self.J[o], self.X[o])

    def to_dict(self):
        return {(int(i), int(j)): x.item() for i, j, x in zip(self.I, self.J, self.X)}


def mxm(Cm, A, B, add, mul, sr_type, mask=None, accum=None, accum_type=None, replace=False, mask_comp=False,
        mask_struct=False, tran_a=False, tran_b=False):
    """C<M,r> = accum(C, op(A) add.mul op(B)); returns the new C as Tuples."""
    L = lib()
    flags = (F_REPLACE if replace else 0) | (F_MASK_COMP if mask_comp else 0) | (F_MASK_STRUCT if mask_struct else 0) | \
        (F_TRAN_A if tran_a else 0) | (F_TRAN_B if tran_b else 0)
    on = C.c_uint64(0)
    oi, oj, ox = C.c_void_p(), C.c_void_p(), C.c_void_p()
    m = mask
    rc = L.oracle_mxm(
        C.c_int(tcode(Cm.typ)), C.c_uint64(Cm.nrows), C.c_uint64(Cm.ncols), C.c_uint64(Cm.nvals), _p(Cm.I), _p(Cm.J), _p(Cm.X),
        C.c_int(tcode(m.typ) if m is not None else -1), C.c_uint64(m.nvals if m is not None else 0),
        _p(m.I) if m is not None else None, _p(m.J) if m is not None else None, _p(m.X) if m is not None else None,
        C.c_int(opcode(accum)), C.c_int(tcode(accum_type or Cm.typ)), C.c_int(opcode(add)), C.c_int(opcode(mul)), C.c_int(tcode(sr_type)),
        C.c_int(tcode(A.typ)), C.c_uint64(A.nrows), C.c_uint64(A.ncols), C.c_uint64(A.nvals), _p(A.I), _p(A.J), _p(A.X),
        C.c_int(tcode(B.typ)), C.c_uint64(B.nrows), C.c_uint64(B.ncols), C.c_uint64(B.nvals), _p(B.I), _p(B.J), _p(B.X),
        C.c_int(flags), C.byref(on), C.byref(oi), C.byref(oj), C.byref(ox))
    if rc:
        raise ValueError(f"oracle_mxm failed with GrB_Info {rc}")
    n = on.value
    dt = NP[Cm.typ]
    I = np.ctypeslib.as_array(C.cast(oi, C.POINTER(C.c_uint64)), (n,)).copy() if n else np.zeros(0, np.uint64)
    J = np.ctypeslib.as_array(C.cast(oj, C.POINTER(C.c_uint64)), (n,)).copy() if n else np.zeros(0, np.uint64)
    X = np.frombuffer(C.string_at(ox, n * np.dtype(dt).itemsize), dtype=dt).copy() if n else np.zeros(0, dt)
    for p in (oi, oj, ox):
        L.oracle_free(p)
    return Tuples(Cm.typ, Cm.nrows, Cm.ncols, I, J, X)


def col_vector(typ, n, idx=(), vals=()):
    idx = _arr(idx, np.uint64)
    return Tuples(typ, n, 1, idx, np.zeros(len(idx), np.uint64), vals)


def row_vector(typ, n, idx=(), vals=()):
    idx = _arr(idx, np.uint64)
    return Tuples(typ, 1, n, np.zeros(len(idx), np.uint64), idx, vals)


def mxv(w, A, u, add, mul, sr_type, mask=None, **kw):
    """w<mask,r> = accum(w, op(A) add.mul u) with w, u, mask column vectors (idx, vals given as Tuples n x 1)."""
    kw.pop("tran_b", None)
    return mxm(w, A, u, add, mul, sr_type, mask=mask, **kw)


def vxm(w, u, A, add, mul, sr_type, mask=None, tran_a=False, **kw):
    """w'<mask',r> = accum(w', u' add.mul op(A)) with w, u, mask row vectors (Tuples 1 x n); tran_a means desc.INP1."""
    return mxm(w, u, A, add, mul, sr_type, mask=mask, tran_b=tran_a, **kw)


def reduce(typ, X, op):
    X = _arr(X, NP[typ])
    out = np.zeros(1, NP[typ])
    lib().oracle_reduce(C.c_int(tcode(typ)), C.c_uint64(len(X)), _p(X), C.c_int(opcode(op)), _p(out))
    return out[0]


# ---- typed fast paths (the timed CPU baseline) -------------------------------------------------
def fast_spmv(rowptr, col, val, x, semiring="PLUS_TIMES"):
    L = lib()
    n = len(rowptr) - 1
    rowptr, col = _arr(rowptr, np.uint32), _arr(col, np.uint32)
    pres = np.zeros(n, np.uint8)
    if semiring == "PLUS_TIMES" and np.asarray(x).dtype == np.float64:
        val, x = _arr(val, np.float64), _arr(x, np.float64); y = np.zeros(n, np.float64)
        L.fast_spmv_plus_times_fp64(C.c_uint32(n), _p(rowptr), _p(col), _p(val), _p(x), _p(y), _p(pres))
    elif semiring == "PLUS_TIMES":
        val, x = _arr(val, np.float32), _arr(x, np.float32); y = np.zeros(n, np.float32)
        L.fast_spmv_plus_times_fp32(C.c_uint32(n), _p(rowptr), _p(col), _p(val), _p(x), _p(y), _p(pres))
    elif semiring == "PLUS_SECOND" and np.asarray(x).dtype == np.float64:
        x = _arr(x, np.float64); y = np.zeros(n, np.float64)
        L.fast_spmv_plus_second_fp64(C.c_uint32(n), _p(rowptr), _p(col), _p(x), _p(y), _p(pres))
    elif semiring == "PLUS_SECOND":
        x = _arr(x, np.float32); y = np.zeros(n, np.float32)
        L.fast_spmv_plus_second_fp32(C.c_uint32(n), _p(rowptr), _p(col), _p(x), _p(y), _p(pres))
    elif semiring == "PLUS_SECOND_WIDE":      # row sums formed in double, rounded to float once
        x = _arr(x, np.float32); y = np.zeros(n, np.float32)
        L.fast_spmv_plus_second_fp32_wide(C.c_uint32(n), _p(rowptr), _p(col), _p(x), _p(y), _p(pres))
    else:
        raise ValueError(semiring)
    return y, pres


def fast_tricount(rowptr, col):
    rowptr, col = _arr(rowptr, np.uint32), _arr(col, np.uint32)
    return int(lib().fast_tricount_LL_maskL(C.c_uint32(len(rowptr) - 1), _p(rowptr), _p(col)))


def fast_masked_mxm(rowptr, col, val=None):
    """Per-entry C<L> = L (+).(x) L on the CSR of L: (out, has) aligned with L's entries — val None: PLUS_PAIR counts (as doubles, exact),
    else PLUS_TIMES on doubles; has[p] = 0 where C has no entry at mask entry p."""
    rowptr, col = _arr(rowptr, np.uint32), _arr(col, np.uint32)
    n = len(rowptr) - 1
    out = np.zeros(len(col), np.float64); has = np.zeros(len(col), np.uint8)
    v = _arr(val, np.float64) if val is not None else None
    lib().fast_masked_mxm_LL(C.c_uint32(n), _p(rowptr), _p(col), _p(v) if v is not None else None, _p(out), _p(has))
    return out, has


def fast_mxm_rows(rowptr, col, val, rows):
    """Rows `rows` of the unmasked C = A (+).(x) A (PLUS_TIMES FP64): (row pointers int64[len(rows)+1], columns uint32, values float64,
    products per row) — columns ascending inside a row, products added in ascending k."""
    rowptr, col, val, rows = _arr(rowptr, np.uint32), _arr(col, np.uint32), _arr(val, np.float64), _arr(rows, np.uint32)
    n, ns = len(rowptr) - 1, len(rows)
    counts = np.zeros(ns, np.int64); prods = np.zeros(ns, np.int64)
    f = lib().fast_mxm_rows_plus_times_fp64
    f(C.c_uint32(n), _p(rowptr), _p(col), _p(val), C.c_uint32(ns), _p(rows), _p(counts), _p(prods), None, None, None)
    off = np.zeros(ns + 1, np.int64); np.cumsum(counts, out=off[1:])
    oc = np.zeros(int(off[-1]), np.uint32); ov = np.zeros(int(off[-1]), np.float64)
    f(C.c_uint32(n), _p(rowptr), _p(col), _p(val), C.c_uint32(ns), _p(rows), None, None, _p(off), _p(oc), _p(ov))
    return off, oc, ov, prods


def fast_bc(rowptr, col, rowptr_t, col_t, sources, max_levels=64):
    """gap/bcmark.py:16-67 on the directed graph A = (rowptr, col), AT = its transpose: (centrality float64[n], depth, frontier sizes)."""
    rowptr, col, rowptr_t, col_t = _arr(rowptr, np.uint32), _arr(col, np.uint32), _arr(rowptr_t, np.uint32), _arr(col_t, np.uint32)
    n = len(rowptr) - 1
    src = _arr(sources, np.uint32)
    cent = np.zeros(n, np.float64); lv = np.zeros(max_levels + 1, np.int64)
    f = lib().fast_bc_batch
    f.restype = C.c_int
    depth = f(C.c_uint32(n), _p(rowptr), _p(col), _p(rowptr_t), _p(col_t), _p(src), C.c_int(len(src)), _p(cent), _p(lv), C.c_int(max_levels))
    return cent, int(depth), lv[:depth].tolist()


def fast_bfs(rowptr, col, src):
    rowptr, col = _arr(rowptr, np.uint32), _arr(col, np.uint32)
    n = len(rowptr) - 1
    lev = np.zeros(n, np.uint8)
    depth = lib().fast_bfs_levels(C.c_uint32(n), _p(rowptr), _p(col), C.c_uint32(src), _p(lev))
    return lev, depth


def fast_sssp(rowptr, col, val, src, max_sweeps=1 << 30):
    """The reference's MIN_PLUS shortest-path loop (v.vxm(A, MIN_PLUS, accum=MIN, out=v) until nothing changes) on a CSR with
    INT64 or FP64 weights: returns (dist, present, sweeps) — `sweeps` counts the products, the last of which changed nothing."""
    rowptr, col = _arr(rowptr, np.uint32), _arr(col, np.uint32)
    n = len(rowptr) - 1
    val = np.ascontiguousarray(val)
    pres = np.zeros(n, np.uint8)
    if val.dtype == np.int64:
        dist = np.zeros(n, np.int64); fn = lib().fast_sssp_min_plus_int64
    elif val.dtype == np.float64:
        dist = np.zeros(n, np.float64); fn = lib().fast_sssp_min_plus_fp64
    else:
        raise ValueError(val.dtype)
    sweeps = fn(C.c_uint32(n), _p(rowptr), _p(col), _p(val), C.c_uint32(src), C.c_int(min(max_sweeps, 1 << 30)), _p(dist), _p(pres))
    return dist, pres, int(sweeps)


def num_threads():
    return lib().oracle_num_threads()
