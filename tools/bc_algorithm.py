"""The batched betweenness centrality of the reference's GAP driver (gap/bcmark.py:16-67), statement for statement over the
pygraphblas_amd mirror.  The driver itself is written against descriptor names the reference no longer has (`oocr`, `Replace`,
`TransposeA`: stale API); they are RC, R and T0 in its current descriptor.py.  Shared by tests and tools."""


def bc(gb, sources, AT, A, sizes=None, typ=None):
    """`sizes` (a list) receives the entry count of every level's frontier (the at-scale parity test compares them with the oracle's).
    `typ`: the arithmetic type (the driver's is FP32; the parity test also runs it in FP64, where 1e-6 against the oracle's doubles is meaningful)."""
    from pygraphblas_amd import descriptor as D
    Matrix, Vector, FP32, BOOL = gb.Matrix, gb.Vector, (typ or gb.FP32), gb.BOOL
    n = A.nrows
    ns = len(sources)
    paths = Matrix.dense(FP32, ns, n, 0)
    frontier = Matrix.sparse(FP32, ns, n)
    S = []
    for i, s in enumerate(sources):
        paths[i, sources[i]] = 1
        frontier[i, sources[i]] = 1
    frontier.mxm(A, out=frontier, mask=paths, semiring=FP32.PLUS_FIRST, desc=D.RC)
    depth = 0
    for depth in range(n):
        if frontier.nvals == 0:
            break
        if sizes is not None:
            sizes.append(frontier.nvals)
        s = Matrix.sparse(BOOL, ns, n)
        frontier.apply(BOOL.ONE, out=s)
        S.append(s)
        paths.assign_matrix(frontier, accum=FP32.PLUS)
        frontier.mxm(A, out=frontier, mask=paths, semiring=FP32.PLUS_FIRST, desc=D.RC)
    bcu = Matrix.dense(FP32, ns, n, 1)
    W = Matrix.sparse(FP32, ns, n)
    for i in range(depth - 1, 0, -1):
        bcu.emult(paths, FP32.DIV, out=W, mask=S[i], desc=D.R)
        W.mxm(AT, out=W, mask=S[i - 1], semiring=FP32.PLUS_FIRST, desc=D.R)
        W.emult(paths, FP32.TIMES, out=bcu, accum=FP32.PLUS)
    centrality = Vector.dense(FP32, n, -ns)
    bcu.reduce_vector(accum=FP32.PLUS, out=centrality, desc=D.T0)
    return centrality, depth
