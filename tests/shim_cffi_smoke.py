"""Run under a Python that has cffi (the image's /opt/conda/bin/python3.9): drives the MI355X backend through the
`suitesparse_graphblas`-compatible CFFI shim exactly the way pygraphblas drives SuiteSparse (ffi.new handles,
lib.GrB_* calls, GrB_Info return codes).  Prints OK and the result; exits non-zero on any mismatch.
Usage: python3.9 tests/shim_cffi_smoke.py [--gpu]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "shim"))
from suitesparse_graphblas import lib, ffi, initialize, is_initialized  # noqa: E402

if not is_initialized():
    initialize(blocking=False, memory_manager="c")


def check(info):
    assert info == lib.GrB_SUCCESS, f"GrB_Info {info}"


names = dir(lib)
assert "GrB_PLUS_TIMES_SEMIRING_INT64" in names and "GxB_PLUS_PAIR_INT64" in names and "GrB_DESC_RC" in names
A = ffi.new("GrB_Matrix*")
check(lib.GrB_Matrix_new(A, lib.GrB_INT64, 3, 3))
for i, j, x in ((0, 1, 1), (1, 2, 2), (2, 0, 3)):
    check(lib.GrB_Matrix_setElement_INT64(A[0], x, i, j))
v = ffi.new("GrB_Vector*")
check(lib.GrB_Vector_new(v, lib.GrB_INT64, 3))
for i, x in ((0, 2), (1, 3), (2, 4)):
    check(lib.GrB_Vector_setElement_INT64(v[0], x, i))
n = ffi.new("GrB_Index*")
check(lib.GrB_Matrix_nvals(n, A[0]))
assert n[0] == 3
zt = ffi.new("GrB_Type*"); mon = ffi.new("GrB_Monoid*"); op = ffi.new("GrB_BinaryOp*")
check(lib.GxB_Semiring_add(mon, lib.GrB_PLUS_TIMES_SEMIRING_INT64)); check(lib.GxB_Monoid_operator(op, mon[0])); check(lib.GxB_BinaryOp_ztype(zt, op[0]))
assert zt[0] == lib.GrB_INT64
w = ffi.new("GrB_Vector*")
check(lib.GrB_Vector_new(w, lib.GrB_INT64, 3))
info = lib.GrB_mxv(w[0], ffi.NULL, ffi.NULL, lib.GrB_PLUS_TIMES_SEMIRING_INT64, A[0], v[0], ffi.NULL)
if "--gpu" in sys.argv:
    check(info)
    I = ffi.new("GrB_Index[3]"); X = ffi.new("int64_t[3]"); n[0] = 3
    check(lib.GrB_Vector_extractTuples_INT64(I, X, n, w[0]))
    assert list(I) == [0, 1, 2] and list(X) == [3, 8, 6], (list(I), list(X))      # reference doctest matrix.py:2610-2616
    C = ffi.new("GrB_Matrix*")
    check(lib.GrB_Matrix_new(C, lib.GrB_INT64, 3, 3))
    check(lib.GrB_mxm(C[0], ffi.NULL, ffi.NULL, lib.GrB_MIN_PLUS_SEMIRING_INT64, A[0], A[0], ffi.NULL))
    r = ffi.new("int64_t*")
    check(lib.GrB_Matrix_reduce_INT64(r, ffi.NULL, lib.GrB_PLUS_MONOID_INT64, C[0], ffi.NULL))
    assert r[0] == (1 + 2) + (2 + 3) + (3 + 1), r[0]
    # ---- the reference's own hot-path tests, restated as the raw CFFI call sequences its methods make --------------------
    def mat(I, J, V, typ="INT64", nr=None, nc=None):
        T = getattr(lib, "GrB_" + typ); h = ffi.new("GrB_Matrix*")
        check(lib.GrB_Matrix_new(h, T, nr or max(I) + 1, nc or max(J) + 1))
        for i, j, x in zip(I, J, V):
            check(getattr(lib, "GrB_Matrix_setElement_" + typ)(h[0], x, i, j))
        return h

    def vec(I, V, typ="INT64", size=None):
        T = getattr(lib, "GrB_" + typ); h = ffi.new("GrB_Vector*")
        check(lib.GrB_Vector_new(h, T, size or max(I) + 1))
        for i, x in zip(I, V):
            check(getattr(lib, "GrB_Vector_setElement_" + typ)(h[0], x, i))
        return h

    CT = {"INT64": "int64_t", "BOOL": "_Bool"}

    def vtuples(h, typ="INT64"):
        nv = ffi.new("GrB_Index*"); check(lib.GrB_Vector_nvals(nv, h[0])); k = nv[0]
        I = ffi.new("GrB_Index[]", max(k, 1)); X = ffi.new(CT[typ] + "[]", max(k, 1))
        check(getattr(lib, "GrB_Vector_extractTuples_" + typ)(I, X, nv, h[0]))
        return list(I)[:k], [int(x) for x in list(X)[:k]]

    def mtuples(h, typ="INT64"):
        nv = ffi.new("GrB_Index*"); check(lib.GrB_Matrix_nvals(nv, h[0])); k = nv[0]
        I = ffi.new("GrB_Index[]", max(k, 1)); J = ffi.new("GrB_Index[]", max(k, 1)); X = ffi.new(CT[typ] + "[]", max(k, 1))
        check(getattr(lib, "GrB_Matrix_extractTuples_" + typ)(I, J, X, nv, h[0]))
        return sorted(zip(list(I)[:k], list(J)[:k], [int(x) for x in list(X)[:k]]))

    # tests/test_matrix.py:249-262 test_mxm (default PLUS_TIMES, then LOR_LAND on the INT64 operands)
    m = mat([0, 1, 2], [1, 2, 0], [1, 2, 3]); nn = mat([0, 1, 2], [1, 2, 0], [2, 3, 4])
    o = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(o, lib.GrB_INT64, 3, 3))
    check(lib.GrB_mxm(o[0], ffi.NULL, ffi.NULL, lib.GrB_PLUS_TIMES_SEMIRING_INT64, m[0], nn[0], ffi.NULL))
    assert mtuples(o) == [(0, 2, 3), (1, 0, 8), (2, 1, 6)], mtuples(o)
    ob = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(ob, lib.GrB_BOOL, 3, 3))
    check(lib.GrB_mxm(ob[0], ffi.NULL, ffi.NULL, lib.GrB_LOR_LAND_SEMIRING_BOOL, m[0], nn[0], ffi.NULL))
    assert mtuples(ob, "BOOL") == [(0, 2, 1), (1, 0, 1), (2, 1, 1)]
    # tests/test_matrix.py:265-277 test_mxm_context: PLUS_PLUS, and the T0 descriptor
    check(lib.GrB_mxm(o[0], ffi.NULL, ffi.NULL, lib.GxB_PLUS_PLUS_INT64, m[0], nn[0], ffi.NULL))
    assert mtuples(o) == [(0, 2, 4), (1, 0, 6), (2, 1, 5)], mtuples(o)
    check(lib.GrB_mxm(o[0], ffi.NULL, ffi.NULL, lib.GrB_PLUS_TIMES_SEMIRING_INT64, m[0], nn[0], lib.GrB_DESC_T0))
    assert mtuples(o) == [(0, 0, 12), (1, 1, 2), (2, 2, 6)], mtuples(o)          # m' n: m'(j,i)=m(i,j)
    # tests/test_matrix.py:292-306 test_mxv (+ transpose with T0, + PLUS_PLUS)
    m4 = mat([0, 1, 2, 3], [1, 2, 0, 1], [1, 2, 3, 4]); v3 = vec([0, 1, 2], [2, 3, 4])
    w4 = ffi.new("GrB_Vector*"); check(lib.GrB_Vector_new(w4, lib.GrB_INT64, 4))
    check(lib.GrB_mxv(w4[0], ffi.NULL, ffi.NULL, lib.GrB_PLUS_TIMES_SEMIRING_INT64, m4[0], v3[0], ffi.NULL))
    assert vtuples(w4) == ([0, 1, 2, 3], [3, 8, 6, 12]), vtuples(w4)
    mt = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(mt, lib.GrB_INT64, 3, 4)); check(lib.GrB_transpose(mt[0], ffi.NULL, ffi.NULL, m4[0], ffi.NULL))
    check(lib.GrB_mxv(w4[0], ffi.NULL, ffi.NULL, lib.GrB_PLUS_TIMES_SEMIRING_INT64, mt[0], v3[0], lib.GrB_DESC_T0))
    assert vtuples(w4) == ([0, 1, 2, 3], [3, 8, 6, 12])
    check(lib.GrB_mxv(w4[0], ffi.NULL, ffi.NULL, lib.GxB_PLUS_PLUS_INT64, m4[0], v3[0], ffi.NULL))
    assert vtuples(w4) == ([0, 1, 2, 3], [4, 6, 5, 7]), vtuples(w4)
    # tests/test_vector.py:298-315 test_vxm (masked, T1 on the transpose, PLUS_PLUS)
    mv = mat([0, 1, 2, 0], [1, 2, 0, 3], [1, 2, 3, 4]); j = vec([1], [True], "BOOL", 4)
    check(lib.GrB_vxm(w4[0], ffi.NULL, ffi.NULL, lib.GrB_PLUS_TIMES_SEMIRING_INT64, v3[0], mv[0], ffi.NULL))
    assert vtuples(w4) == ([0, 1, 2, 3], [12, 2, 6, 8]), vtuples(w4)
    l = ffi.new("GrB_Vector*"); check(lib.GrB_Vector_new(l, lib.GrB_INT64, 4))
    check(lib.GrB_vxm(l[0], j[0], ffi.NULL, lib.GrB_PLUS_TIMES_SEMIRING_INT64, v3[0], mv[0], ffi.NULL))
    assert vtuples(l) == ([1], [2]), vtuples(l)
    mvt = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(mvt, lib.GrB_INT64, 4, 3)); check(lib.GrB_transpose(mvt[0], ffi.NULL, ffi.NULL, mv[0], ffi.NULL))
    check(lib.GrB_vxm(w4[0], ffi.NULL, ffi.NULL, lib.GrB_PLUS_TIMES_SEMIRING_INT64, v3[0], mvt[0], lib.GrB_DESC_T1))
    assert vtuples(w4) == ([0, 1, 2, 3], [12, 2, 6, 8])
    check(lib.GrB_vxm(w4[0], ffi.NULL, ffi.NULL, lib.GxB_PLUS_PLUS_INT64, v3[0], mv[0], ffi.NULL))
    assert vtuples(w4) == ([0, 1, 2, 3], [7, 3, 5, 6]), vtuples(w4)
    # tests/test_descriptor.py:13-30 test_RCT0 / test_RC: output aliases the operand, complemented empty mask, replace
    for desc, want in ((lib.GrB_DESC_RCT0, [1]), (lib.GrB_DESC_RC, [2])):
        Mb = mat([0, 1, 2], [1, 2, 0], [True, True, True], "BOOL")
        wb = vec([0], [True], "BOOL", 3); vb = ffi.new("GrB_Vector*"); check(lib.GrB_Vector_new(vb, lib.GrB_BOOL, 3))
        check(lib.GrB_mxv(wb[0], vb[0], ffi.NULL, lib.GrB_LOR_LAND_SEMIRING_BOOL, Mb[0], wb[0], desc))
        assert vtuples(wb, "BOOL") == (want, [1]), vtuples(wb, "BOOL")
    print("OK gpu", list(X), r[0], "+ reference test_mxm/test_mxm_context/test_mxv/test_vxm/test_RCT0/test_RC sequences")
else:
    assert info == lib.GrB_PANIC, info                                                # no device: fail loudly
    err = ffi.new("char**"); check(lib.GrB_Vector_error(err, w[0]))
    assert b"no device" in ffi.string(err[0])
    print("OK cpu (compute refused without a device)")
for h, fn in ((A, lib.GrB_Matrix_free), (v, lib.GrB_Vector_free), (w, lib.GrB_Vector_free)):
    check(fn(h)); check(fn(h))                                                        # double free is a no-op
