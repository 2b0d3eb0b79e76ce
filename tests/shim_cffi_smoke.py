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
    # tests/test_matrix.py:858-864 test_pow: (m @ m) on a dense UINT8 matrix wraps modulo 256
    import random
    rnd = random.Random(4); D8 = [[rnd.randrange(256) for _ in range(6)] for _ in range(6)]
    m8 = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(m8, lib.GrB_UINT8, 6, 6))
    for i in range(6):
        for jj in range(6):
            check(lib.GrB_Matrix_setElement_UINT8(m8[0], D8[i][jj], i, jj))
    p8 = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(p8, lib.GrB_UINT8, 6, 6))
    check(lib.GrB_mxm(p8[0], ffi.NULL, ffi.NULL, lib.GrB_PLUS_TIMES_SEMIRING_UINT8, m8[0], m8[0], ffi.NULL))
    nv8 = ffi.new("GrB_Index*"); check(lib.GrB_Matrix_nvals(nv8, p8[0])); assert nv8[0] == 36
    I8 = ffi.new("GrB_Index[36]"); J8 = ffi.new("GrB_Index[36]"); X8 = ffi.new("uint8_t[36]")
    check(lib.GrB_Matrix_extractTuples_UINT8(I8, J8, X8, nv8, p8[0]))
    for q in range(36):
        assert X8[q] == sum(D8[I8[q]][k] * D8[k][J8[q]] for k in range(6)) % 256
    # ---- the reference's LOOPS as the raw call sequences its methods make, in the non-blocking mode the reference initialises ----
    # (the library defers and fuses the vector operations between the observations: results must be as-if sequential)
    rnd = random.Random(11); NV = 60
    edges = sorted({(rnd.randrange(NV), rnd.randrange(NV)) for _ in range(400)})
    Ag = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(Ag, lib.GrB_FP32, NV, NV))
    for (i, jj) in edges:
        check(lib.GrB_Matrix_setElement_FP32(Ag[0], 1.0, i, jj))
    outdeg = [0] * NV
    for (i, jj) in edges:
        outdeg[i] += 1

    def fvec(n):
        h = ffi.new("GrB_Vector*"); check(lib.GrB_Vector_new(h, lib.GrB_FP32, n)); return h

    # gap/prmark.py:8-30
    dv = fvec(NV)
    for i in range(NV):
        if outdeg[i]:
            check(lib.GrB_Vector_setElement_FP32(dv[0], float(outdeg[i]), i))
    damping = 0.85
    r_ = fvec(NV); t_ = fvec(NV)
    check(lib.GrB_Vector_assign_FP32(dv[0], ffi.NULL, lib.GrB_DIV_FP32, damping, lib.GrB_ALL, NV, ffi.NULL))
    check(lib.GrB_Vector_assign_FP32(r_[0], ffi.NULL, ffi.NULL, 1.0 / NV, lib.GrB_ALL, NV, ffi.NULL))
    teleport = (1 - damping) / NV
    rr = [1.0 / NV] * NV; tt = [0.0] * NV; its = 0; rdiff = ffi.new("float*")
    for it in range(100):
        t_, r_ = r_, t_
        wv = fvec(NV)
        check(lib.GrB_Vector_eWiseMult_BinaryOp(wv[0], ffi.NULL, ffi.NULL, lib.GrB_DIV_FP32, t_[0], dv[0], ffi.NULL))                  # w = t / d
        check(lib.GrB_Vector_assign_FP32(r_[0], ffi.NULL, ffi.NULL, teleport, lib.GrB_ALL, NV, ffi.NULL))                             # r[:] = teleport
        check(lib.GrB_mxv(r_[0], ffi.NULL, lib.GrB_PLUS_FP32, lib.GxB_PLUS_SECOND_FP32, Ag[0], wv[0], lib.GrB_DESC_T0))              # r += A' (+).second w
        check(lib.GrB_Vector_eWiseAdd_BinaryOp(t_[0], ffi.NULL, ffi.NULL, lib.GrB_MINUS_FP32, r_[0], t_[0], ffi.NULL))                # t -= r  (the reference's operand order)
        check(lib.GrB_Vector_apply(t_[0], ffi.NULL, ffi.NULL, lib.GrB_ABS_FP32, t_[0], ffi.NULL))                                    # t = abs(t)
        check(lib.GrB_Vector_reduce_FP32(rdiff, ffi.NULL, lib.GrB_PLUS_MONOID_FP32, t_[0], ffi.NULL))                                # rdiff = t.reduce_float()
        check(lib.GrB_Vector_free(wv))
        # the same iteration in Python floats
        tt, rr = rr, tt
        w_ = [tt[i] / (outdeg[i] / damping) if outdeg[i] else None for i in range(NV)]
        rr = [teleport] * NV
        for (i, jj) in edges:
            if w_[i] is not None:
                rr[jj] += w_[i]
        ref_rdiff = sum(abs(a - b) for a, b in zip(tt, rr))
        its += 1
        assert abs(rdiff[0] - ref_rdiff) <= 1e-5 * max(ref_rdiff, 1e-6) + 1e-7, (it, rdiff[0], ref_rdiff)
        if rdiff[0] <= 1e-4:
            break
    nvr = ffi.new("GrB_Index*"); check(lib.GrB_Vector_nvals(nvr, r_[0])); assert nvr[0] == NV
    xr = ffi.new("float*")
    for i in range(NV):
        check(lib.GrB_Vector_extractElement_FP32(xr, r_[0], i)); assert abs(xr[0] - rr[i]) <= 1e-5 * abs(rr[i]) + 1e-9, (i, xr[0], rr[i])
    # demo/Introduction-to-GraphBLAS-with-Python.ipynb:4301-4313 (BFS) on the same graph as BOOL
    Ab = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(Ab, lib.GrB_BOOL, NV, NV))
    for (i, jj) in edges:
        check(lib.GrB_Matrix_setElement_BOOL(Ab[0], True, i, jj))
    lv = ffi.new("GrB_Vector*"); check(lib.GrB_Vector_new(lv, lib.GrB_UINT8, NV))
    qv = ffi.new("GrB_Vector*"); check(lib.GrB_Vector_new(qv, lib.GrB_BOOL, NV)); check(lib.GrB_Vector_setElement_BOOL(qv[0], True, edges[0][0]))
    go = ffi.new("_Bool*"); level = 1
    while True:
        check(lib.GrB_Vector_reduce_BOOL(go, ffi.NULL, lib.GrB_LOR_MONOID_BOOL, qv[0], ffi.NULL))
        if not go[0] or level > NV:
            break
        check(lib.GrB_Vector_assign_UINT8(lv[0], qv[0], ffi.NULL, level, lib.GrB_ALL, NV, ffi.NULL))
        check(lib.GrB_vxm(qv[0], lv[0], ffi.NULL, lib.GrB_LOR_LAND_SEMIRING_BOOL, lv[0], Ab[0], lib.GrB_DESC_RC))
        level += 1
    lev = [0] * NV; lev[edges[0][0]] = 1; frontier = [edges[0][0]]; d_ = 1
    while frontier:
        nxt = sorted({jj for (i, jj) in edges if i in frontier and not lev[jj]}); d_ += 1
        for jj in nxt:
            lev[jj] = d_
        frontier = nxt
    u8 = ffi.new("uint8_t*")
    for i in range(NV):
        info = lib.GrB_Vector_extractElement_UINT8(u8, lv[0], i)
        assert (info == lib.GrB_NO_VALUE and lev[i] == 0) or (info == lib.GrB_SUCCESS and u8[0] == lev[i]), (i, info, u8[0], lev[i])
    # a complemented mask without a mask object allows no writes: `w(:) += u` under GrB_DESC_C leaves w as it is (the whole-vector
    # and whole-matrix accumulate fast paths used to forward to eWiseAdd without the descriptor)
    wa, ua = vec([0, 2], [10, 20], size=4), vec([0, 1], [1, 2], size=4)
    check(lib.GrB_Vector_assign(wa[0], ffi.NULL, lib.GrB_PLUS_INT64, ua[0], lib.GrB_ALL, 4, lib.GrB_DESC_C))
    assert vtuples(wa) == ([0, 2], [10, 20]), vtuples(wa)
    check(lib.GrB_Vector_assign(wa[0], ffi.NULL, lib.GrB_PLUS_INT64, ua[0], lib.GrB_ALL, 4, ffi.NULL))
    assert vtuples(wa) == ([0, 1, 2], [11, 2, 20]), vtuples(wa)
    Ca, Aa = mat([0, 1], [0, 1], [10, 20], nr=2, nc=2), mat([0, 0], [0, 1], [1, 2], nr=2, nc=2)
    check(lib.GrB_Matrix_assign(Ca[0], ffi.NULL, lib.GrB_PLUS_INT64, Aa[0], lib.GrB_ALL, 2, lib.GrB_ALL, 2, lib.GrB_DESC_C))
    assert mtuples(Ca) == [(0, 0, 10), (1, 1, 20)], mtuples(Ca)
    check(lib.GrB_Matrix_assign(Ca[0], ffi.NULL, lib.GrB_PLUS_INT64, Aa[0], lib.GrB_ALL, 2, lib.GrB_ALL, 2, ffi.NULL))
    assert mtuples(Ca) == [(0, 0, 11), (0, 1, 2), (1, 1, 20)], mtuples(Ca)
    print("OK gpu", list(X), r[0], "+ reference test_mxm/test_mxm_context/test_mxv/test_vxm/test_RCT0/test_RC/test_pow sequences + the PageRank (%d iterations) and BFS (%d levels) loops in non-blocking mode" % (its, level - 1))
else:
    assert info == lib.GrB_PANIC, info                                                # no device: fail loudly
    err = ffi.new("char**"); check(lib.GrB_Vector_error(err, w[0]))
    assert b"no device" in ffi.string(err[0])
    print("OK cpu (compute refused without a device)")
# ---- container features outside the device layouts (host mirror; same answers with or without a GPU) ------------------
IMAX = lib.GxB_INDEX_MAX
H = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(H, lib.GrB_INT8, IMAX, IMAX))           # hypersparse default dims (matrix.py:167-170)
check(lib.GrB_Matrix_setElement_INT8(H[0], 42, 0, 1)); check(lib.GrB_Matrix_setElement_INT8(H[0], 42, 0, 2))
Mk = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(Mk, lib.GrB_BOOL, IMAX, IMAX)); check(lib.GrB_Matrix_setElement_BOOL(Mk[0], True, 1, 1))
F = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(F, lib.GrB_FP64, IMAX, IMAX))
check(lib.GrB_Matrix_assign_FP64(F[0], Mk[0], ffi.NULL, 3.14, lib.GrB_ALL, 0, lib.GrB_ALL, 0, ffi.NULL))          # Matrix.sparse(float, fill=3.14, mask=mask)
x = ffi.new("double*"); check(lib.GrB_Matrix_extractElement_FP64(x, F[0], 1, 1)); check(lib.GrB_Matrix_nvals(n, F[0])); assert x[0] == 3.14 and n[0] == 1
iso = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(iso, lib.GrB_INT64, IMAX, IMAX))
check(lib.GrB_Matrix_assign_INT64(iso[0], ffi.NULL, ffi.NULL, 3, lib.GrB_ALL, 0, lib.GrB_ALL, 0, ffi.NULL))       # Matrix.iso(3): readable element-wise
y = ffi.new("int64_t*"); check(lib.GrB_Matrix_extractElement_INT64(y, iso[0], 42, 42)); assert y[0] == 3
assert lib.GrB_Matrix_setElement_INT64(iso[0], 1, 0, 0) == lib.GrB_INSUFFICIENT_SPACE
dg = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(dg, lib.GrB_INT64, 4, 4))
check(lib.GxB_Matrix_diag(dg[0], v[0], -1, ffi.NULL))                                                                # Matrix.from_diag(v, -1), matrix.py:360-366
I3 = ffi.new("GrB_Index[3]"); J3 = ffi.new("GrB_Index[3]"); X3 = ffi.new("int64_t[3]"); n[0] = 3
check(lib.GrB_Matrix_extractTuples_INT64(I3, J3, X3, n, dg[0])); assert (list(I3), list(J3), list(X3)) == ([1, 2, 3], [0, 1, 2], [2, 3, 4])
dv = ffi.new("GrB_Vector*"); check(lib.GrB_Vector_new(dv, lib.GrB_INT64, 3)); check(lib.GxB_Vector_diag(dv[0], dg[0], -1, ffi.NULL))
check(lib.GrB_Vector_extractTuples_INT64(I3, X3, n, dv[0])); assert (list(I3), list(X3)) == ([0, 1, 2], [2, 3, 4])
assert lib.GxB_Vector_diag(dv[0], dg[0], 2, ffi.NULL) == lib.GrB_DIMENSION_MISMATCH                                  # that diagonal has 2 positions
Z = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(Z, lib.GxB_FC64, 2, 2)); check(lib.GxB_Matrix_setElement_FC64(Z[0], 3 + 4j, 0, 1))
z = ffi.new("double _Complex*"); check(lib.GxB_Matrix_extractElement_FC64(z, Z[0], 0, 1)); assert z[0] == 3 + 4j
Z2 = ffi.new("GrB_Matrix*"); check(lib.GrB_Matrix_new(Z2, lib.GxB_FC64, 2, 2))
assert lib.GrB_transpose(Z2[0], ffi.NULL, ffi.NULL, Z[0], ffi.NULL) in (lib.GrB_DOMAIN_MISMATCH, lib.GrB_PANIC)    # no arithmetic on complex containers
if "--gpu" in sys.argv:
    r8 = ffi.new("int64_t*"); check(lib.GrB_Matrix_reduce_INT64(r8, ffi.NULL, lib.GrB_PLUS_MONOID_INT64, H[0], ffi.NULL)); assert r8[0] == 84   # matrix.py:1785-1791
    check(lib.GrB_Matrix_reduce_INT64(r8, ffi.NULL, lib.GrB_MIN_MONOID_INT8, H[0], ffi.NULL)); assert r8[0] == 42
    assert lib.GrB_transpose(Z2[0], ffi.NULL, ffi.NULL, Z[0], ffi.NULL) == lib.GrB_DOMAIN_MISMATCH
    fv = ffi.new("GrB_Vector*"); check(lib.GrB_Vector_new(fv, lib.GrB_FP32, 10))
    vals = (0.6394267678260803, 0.025010755658149719, 0.27502930164337158)
    for i, xv in zip((1, 4, 8), vals):
        check(lib.GrB_Vector_setElement_FP32(fv[0], xv, i))
    d = ffi.new("double*"); check(lib.GrB_Vector_reduce_FP64(d, ffi.NULL, lib.GrB_PLUS_MONOID_FP32, fv[0], ffi.NULL))
    import struct
    f32 = lambda q: struct.unpack("f", struct.pack("f", q))[0]
    assert d[0] == f32(f32(f32(vals[0]) + f32(vals[1])) + f32(vals[2])), d[0]     # a few values fold in index order (the sequential association)
    print("OK gpu containers: hypersparse reduce, complex refusal, sequential small reduce")
print("OK containers: hypersparse masked assign, iso, diag, complex entries")
for h, fn in ((A, lib.GrB_Matrix_free), (v, lib.GrB_Vector_free), (w, lib.GrB_Vector_free)):
    check(fn(h)); check(fn(h))                                                        # double free is a no-op
