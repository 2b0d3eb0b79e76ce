"""The reference's own driver loops around the hot path, restated over the mirror classes — what bench.py times and the
parity tests check at the BASELINE.json sizes.  Each is the reference's code line for line (same calls, same operators,
same stopping rule); nothing here is part of the GraphBLAS API surface.

  pagerank(A, d, ...)      gap/prmark.py:8-30          FP32 PLUS_SECOND, accum PLUS, desc T0          (configs[4])
  bfs(A, start)            demo/Introduction-to-GraphBLAS-with-Python.ipynb:4301-4313  BOOL LOR_LAND  (configs[2])
  sssp(A, start)           demo/Intro-Prez.ipynb:1034-1045; pygraphblas/vector.py:883-885  MIN_PLUS, accum MIN
  triangle_count(L)        demo/TriangleCentrality.ipynb:1446-1449  PLUS_PAIR, mask L                 (configs[3])
"""
from . import Vector, UINT8, BOOL, FP32, INT64, descriptor as D, last_kernel_plan      # (the package is initialised by the time this submodule is asked for)


def pagerank(A, d, damping=0.85, itermax=100, tol=1e-4, fixed_iterations=None, trace=None):
    """gap/prmark.py:8-30.  A: adjacency matrix (any type; the semiring ignores its values), d: FP32 out-degrees with no entry
    for dangling vertices (`A.reduce_vector()`); the product runs on A' through the descriptor, as the reference's does.
    `fixed_iterations` (bench.py) runs exactly that many iterations.  Returns (r, iterations, rdiff)."""
    n = A.nrows
    r = Vector.sparse(FP32, n)
    t = Vector.sparse(FP32, n)
    d.assign_scalar(damping, accum=FP32.DIV)
    r[:] = 1.0 / n
    teleport = (1 - damping) / n
    rdiff, its = 1.0, 0
    for i in range(fixed_iterations if fixed_iterations is not None else itermax):
        temp = t; t = r; r = temp
        w = t / d
        r[:] = teleport
        A.mxv(w, out=r, accum=FP32.PLUS, semiring=FP32.PLUS_SECOND, desc=D.T0)
        t -= r
        t.apply(FP32.ABS, out=t)
        rdiff = t.reduce_float()
        its = i + 1
        if trace is not None:
            trace.append(rdiff)
        if fixed_iterations is None and rdiff <= tol:
            break
    return r, its, rdiff


def bfs(A, start, plans=None):
    """Level BFS: levels start at 1, unreached vertices have no entry.  Returns (v, depth)."""
    v = Vector.sparse(UINT8, A.nrows)
    q = Vector.sparse(BOOL, A.nrows)
    q[start] = True
    level = 1
    while q.reduce_bool() and level <= A.nrows:
        v.assign_scalar(level, mask=q)
        v.vxm(A, mask=v, out=q, desc=D.RC)
        if plans is not None:
            plans.append(last_kernel_plan().split("<")[0])
        level += 1
    return v, level - 1


def sssp(A, start, plans=None, max_sweeps=None, before_sweep=None):
    """Shortest path lengths from `start` by repeated `v<accum MIN> = v MIN_PLUS A` until a sweep changes nothing.
    `before_sweep(v)` (bench.py's byte accounting) sees the operand of every product.  Returns (v, sweeps)."""
    typ = A.type
    v = Vector.sparse(typ, A.nrows)
    v[start] = 0
    sweeps = 0
    while max_sweeps is None or sweeps < max_sweeps:
        if before_sweep is not None:
            before_sweep(v)
        w = v.dup()
        v.vxm(A, semiring=typ.MIN_PLUS, accum=typ.MIN, out=v)
        if plans is not None:
            plans.append(last_kernel_plan().split("<")[0])
        sweeps += 1
        if w.iseq(v):
            break
    return v, sweeps


def triangle_count(L):
    return L.mxm(L, semiring=INT64.PLUS_PAIR, mask=L).reduce_int()
