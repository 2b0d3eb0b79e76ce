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
    print("OK gpu", list(X), r[0])
else:
    assert info == lib.GrB_PANIC, info                                                # no device: fail loudly
    err = ffi.new("char**"); check(lib.GrB_Vector_error(err, w[0]))
    assert b"no device" in ffi.string(err[0])
    print("OK cpu (compute refused without a device)")
for h, fn in ((A, lib.GrB_Matrix_free), (v, lib.GrB_Vector_free), (w, lib.GrB_Vector_free)):
    check(fn(h)); check(fn(h))                                                        # double free is a no-op
