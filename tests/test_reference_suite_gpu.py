"""The reference's own hot-path tests, run against the HIP backend through the mirrored Python surface.

Part 1 replays tests/golden/reference_vectors.json through pygraphblas_amd.  Part 2 restates the reference test
functions for this path almost verbatim (tests/test_matrix.py:249-306, tests/test_vector.py:298-315,
tests/test_descriptor.py:13-30) so they read like the reference's tests: only the import line differs.
"""
import pytest

from golden_runner import load
import golden_runner

pytestmark = pytest.mark.gpu
DATA = load()


@pytest.mark.parametrize("case", DATA["cases"], ids=[c["cite"] for c in DATA["cases"]])
def test_product_matches_reference_vector(gpu, gb, case):
    got = golden_runner.run_product(case, gb)
    assert got == case["expect"], f"{case['cite']}: HIP backend gives {got}, reference pins {case['expect']} [{gb.last_kernel_plan()}]"


def test_mxm(gpu):
    from pygraphblas_amd import Matrix, BOOL
    m = Matrix.from_lists([0, 1, 2], [1, 2, 0], [1, 2, 3])
    n = Matrix.from_lists([0, 1, 2], [1, 2, 0], [2, 3, 4])
    o = m.mxm(n)
    assert o.nrows == 3
    assert o.ncols == 3
    assert o.nvals == 3
    r = Matrix.from_lists([0, 1, 2], [2, 0, 1], [3, 8, 6])
    assert o.iseq(r)
    assert r.iseq(m @ n)
    m @= n
    assert r.iseq(m)
    o = m.mxm(n, semiring=BOOL.LOR_LAND)
    assert o.iseq(Matrix.from_lists([0, 1, 2], [0, 1, 2], [True, True, True]))


def test_mxm_context(gpu):
    from pygraphblas_amd import Matrix, BOOL, INT64, descriptor
    m = Matrix.from_lists([0, 1, 2], [1, 2, 0], [1, 2, 3])
    n = Matrix.from_lists([0, 1, 2], [1, 2, 0], [2, 3, 4])
    with INT64.PLUS_PLUS:
        o = m @ n
    assert o.iseq(Matrix.from_lists([0, 1, 2], [2, 0, 1], [4, 6, 5]))
    with BOOL.LOR_LAND:
        o = m @ n
    assert o.iseq(Matrix.from_lists([0, 1, 2], [2, 0, 1], [True, True, True]))
    with descriptor.T0:
        o = m @ n
    assert o.iseq(m.mxm(n, desc=descriptor.T0))
    with pytest.raises(TypeError):
        m @ 3


def test_mxv(gpu):
    from pygraphblas_amd import Matrix, Vector, INT64, descriptor
    m = Matrix.from_lists([0, 1, 2, 3], [1, 2, 0, 1], [1, 2, 3, 4])
    v = Vector.from_lists([0, 1, 2], [2, 3, 4])
    o = m.mxv(v)
    assert o.iseq(Vector.from_lists([0, 1, 2, 3], [3, 8, 6, 12]))
    assert o.iseq(m @ v)
    assert o.iseq(m.transpose().mxv(v, desc=descriptor.T0))
    with INT64.PLUS_PLUS:
        o = m.mxv(v)
        assert o.iseq(Vector.from_lists([0, 1, 2, 3], [4, 6, 5, 7]))
        assert o.iseq(m @ v)


def test_vxm(gpu):
    from pygraphblas_amd import Matrix, Vector, INT64, descriptor
    m = Matrix.from_lists([0, 1, 2, 0], [1, 2, 0, 3], [1, 2, 3, 4])
    v = Vector.from_lists([0, 1, 2], [2, 3, 4])
    j = Vector.from_lists([1], [True], size=4)
    o = v.vxm(m)
    assert o.iseq(Vector.from_lists([0, 1, 2, 3], [12, 2, 6, 8]))
    l = v.vxm(m, mask=j)
    assert l.iseq(Vector.from_lists([1], [2], size=4))
    assert (v @ m).iseq(o)
    assert v.vxm(m.transpose(), desc=descriptor.T1).iseq(o)
    with INT64.PLUS_PLUS:
        o = v.vxm(m)
        assert o.iseq(Vector.from_lists([0, 1, 2, 3], [7, 3, 5, 6]))
        assert o.iseq(v @ m)


def test_RCT0(gpu):
    from pygraphblas_amd import Matrix, Vector, BOOL, descriptor
    M = Matrix.from_lists([0, 1, 2], [1, 2, 0], [True, True, True])
    w = Vector.sparse(BOOL, 3)
    v = Vector.sparse(BOOL, 3)
    w[0] = True
    M.mxv(w, out=w, mask=v, desc=descriptor.RCT0)
    assert w.iseq(Vector.from_lists([1], [True], 3))


def test_RC(gpu):
    from pygraphblas_amd import Matrix, Vector, BOOL, descriptor
    M = Matrix.from_lists([0, 1, 2], [1, 2, 0], [True, True, True])
    w = Vector.sparse(BOOL, 3)
    v = Vector.sparse(BOOL, 3)
    w[0] = True
    M.mxv(w, out=w, mask=v, desc=descriptor.RC)
    assert w.iseq(Vector.from_lists([2], [True], 3))


def test_promotion(gpu):
    from pygraphblas_amd import Matrix, FP32, FP64, UINT8, INT8
    for case in DATA["promotion"]:
        T = {"FP32": FP32, "FP64": FP64, "UINT8": UINT8, "INT8": INT8}
        m = Matrix.from_lists([0, 1, 2], [1, 2, 0], [1, 2, 3], typ=T[case["left"]])
        n = Matrix.from_lists([0, 1, 2], [1, 2, 0], [2, 3, 4], typ=T[case["right"]])
        assert (m @ n).type is T[case["result"]], case


def test_bfs_reference_loop(gpu):
    """The BFS of demo/Introduction-to-GraphBLAS-with-Python.ipynb cell 31, unchanged, on a small digraph."""
    from pygraphblas_amd import Matrix, Vector, UINT8, BOOL, descriptor

    def bfs(matrix, start):
        v = Vector.sparse(UINT8, matrix.nrows)
        q = Vector.sparse(BOOL, matrix.nrows)
        q[start] = True
        level = 1
        while q.reduce_bool() and level <= matrix.nrows:
            v.assign_scalar(level, mask=q)
            v.vxm(matrix, mask=v, out=q, desc=descriptor.RC)
            level += 1
        return v

    A = Matrix.from_lists([0, 0, 1, 3, 3, 4, 1, 5, 2], [1, 3, 4, 4, 2, 5, 6, 2, 6], [True] * 9, 7, 7)
    assert bfs(A, 0).to_lists() == [[0, 1, 2, 3, 4, 5, 6], [1, 2, 3, 2, 3, 4, 3]]
