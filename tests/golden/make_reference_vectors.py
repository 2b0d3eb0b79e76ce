#!/usr/bin/env python3
"""Writes tests/golden/reference_vectors.json — the golden vectors the reference's own tests and doctests
hold for the mxm / mxv / vxm path.

The reference cannot be executed in this container (its arithmetic lives in SuiteSparse:GraphBLAS, which is
not installed and not under /root/reference), so these vectors are TRANSCRIBED from the reference's test
sources; every case cites the file:line it was read from.  Format of a case:
  op: mxm | mxv | vxm;  A/B: [I, J, X, nrows, ncols];  u/w/mask: [I, X, size];  semiring: "ADD_MUL"; type: semiring type;
  desc: subset of "RSCT0T1";  accum: op name or null;  expect: [I, (J,) X]  (+ expect_type)
"""
import json
import os

m3 = [[0, 1, 2], [1, 2, 0], [1, 2, 3], 3, 3]          # "m" of tests/test_matrix.py:250 and the mxm/mxv/vxm doctests
n3 = [[0, 1, 2], [1, 2, 0], [2, 3, 4], 3, 3]          # "n" of tests/test_matrix.py:251
v3 = [[0, 1, 2], [2, 3, 4], 3]

cases = [
    # ---- tests/test_matrix.py:249-262 test_mxm
    dict(cite="tests/test_matrix.py:249-257", op="mxm", A=m3, B=n3, type="INT64", semiring="PLUS_TIMES", expect=[[0, 1, 2], [2, 0, 1], [3, 8, 6]]),
    # o = m.mxm(n, semiring=BOOL.LOR_LAND) with m already overwritten by m @= n  ->  m = r = [[0,1,2],[2,0,1],[3,8,6]]
    dict(cite="tests/test_matrix.py:259-262", op="mxm", A=[[0, 1, 2], [2, 0, 1], [3, 8, 6], 3, 3], B=n3, type="BOOL", semiring="LOR_LAND",
         expect=[[0, 1, 2], [0, 1, 2], [True, True, True]], expect_type="BOOL"),
    # ---- tests/test_matrix.py:265-290 test_mxm_context
    dict(cite="tests/test_matrix.py:269-272", op="mxm", A=m3, B=n3, type="INT64", semiring="PLUS_PLUS", expect=[[0, 1, 2], [2, 0, 1], [4, 6, 5]]),
    dict(cite="tests/test_matrix.py:274-276", op="mxm", A=m3, B=n3, type="BOOL", semiring="LOR_LAND", expect=[[0, 1, 2], [2, 0, 1], [True, True, True]], expect_type="BOOL"),
    # ---- doctests pygraphblas/matrix.py:2421-2551 (mxm)
    dict(cite="pygraphblas/matrix.py:2462-2471 (accum=INT64.min, out=o=m.dup())", op="mxm", A=m3, B=n3, C=m3, type="INT64", semiring="PLUS_TIMES", accum="MIN",
         expect=[[0, 0, 1, 1, 2, 2], [1, 2, 0, 2, 0, 1], [1, 3, 8, 2, 3, 6]]),
    dict(cite="pygraphblas/matrix.py:2487-2494 (INT64.min_plus)", op="mxm", A=m3, B=n3, type="INT64", semiring="MIN_PLUS", expect=[[0, 1, 2], [2, 0, 1], [4, 6, 5]]),
    dict(cite="pygraphblas/matrix.py:2525-2532 (desc=T0)", op="mxm", A=m3, B=n3, type="INT64", semiring="PLUS_TIMES", desc="T0", expect=[[0, 1, 2], [0, 1, 2], [12, 2, 6]]),
    dict(cite="pygraphblas/matrix.py:2545-2551 (cast=FP32)", op="mxm", A=m3, B=n3, type="INT64", semiring="PLUS_TIMES", expect=[[0, 1, 2], [2, 0, 1], [3.0, 8.0, 6.0]], expect_type="FP32"),
    # ---- tests/test_matrix.py:293-306 test_mxv
    dict(cite="tests/test_matrix.py:293-297", op="mxv", A=[[0, 1, 2, 3], [1, 2, 0, 1], [1, 2, 3, 4], 4, 3], u=v3, type="INT64", semiring="PLUS_TIMES", expect=[[0, 1, 2, 3], [3, 8, 6, 12]]),
    dict(cite="tests/test_matrix.py:301 (transpose + desc=T0)", op="mxv", A=[[1, 2, 0, 1], [0, 1, 2, 3], [1, 2, 3, 4], 3, 4], u=v3, type="INT64", semiring="PLUS_TIMES", desc="T0",
         expect=[[0, 1, 2, 3], [3, 8, 6, 12]]),
    dict(cite="tests/test_matrix.py:303-306 (INT64.PLUS_PLUS)", op="mxv", A=[[0, 1, 2, 3], [1, 2, 0, 1], [1, 2, 3, 4], 4, 3], u=v3, type="INT64", semiring="PLUS_PLUS", expect=[[0, 1, 2, 3], [4, 6, 5, 7]]),
    # ---- doctests pygraphblas/matrix.py:2607-2689 (mxv)
    dict(cite="pygraphblas/matrix.py:2610-2616", op="mxv", A=m3, u=v3, type="INT64", semiring="PLUS_TIMES", expect=[[0, 1, 2], [3, 8, 6]]),
    dict(cite="pygraphblas/matrix.py:2629-2635 (accum=INT64.plus, out=o=v.dup())", op="mxv", A=m3, u=v3, w=v3, type="INT64", semiring="PLUS_TIMES", accum="PLUS", expect=[[0, 1, 2], [5, 11, 10]]),
    dict(cite="pygraphblas/matrix.py:2644-2648 (INT64.min_plus)", op="mxv", A=m3, u=v3, type="INT64", semiring="MIN_PLUS", expect=[[0, 1, 2], [4, 6, 5]]),
    dict(cite="pygraphblas/matrix.py:2666-2670 (desc=T0)", op="mxv", A=m3, u=v3, type="INT64", semiring="PLUS_TIMES", desc="T0", expect=[[0, 1, 2], [12, 2, 6]]),
    dict(cite="pygraphblas/matrix.py:2678-2683 (mask = result of desc=T0 with o[1] deleted)", op="mxv", A=m3, u=v3, mask=[[0, 2], [12, 6], 3], type="INT64", semiring="PLUS_TIMES", expect=[[0, 2], [3, 6]]),
    dict(cite="pygraphblas/matrix.py:2685-2689 (cast=FP32)", op="mxv", A=m3, u=v3, type="INT64", semiring="PLUS_TIMES", expect=[[0, 1, 2], [3.0, 8.0, 6.0]], expect_type="FP32"),
    # ---- tests/test_vector.py:298-315 test_vxm
    dict(cite="tests/test_vector.py:298-303", op="vxm", A=[[0, 1, 2, 0], [1, 2, 0, 3], [1, 2, 3, 4], 3, 4], u=v3, type="INT64", semiring="PLUS_TIMES", expect=[[0, 1, 2, 3], [12, 2, 6, 8]]),
    dict(cite="tests/test_vector.py:305-306 (mask=j)", op="vxm", A=[[0, 1, 2, 0], [1, 2, 0, 3], [1, 2, 3, 4], 3, 4], u=v3, mask=[[1], [True], 4], mask_type="BOOL", type="INT64", semiring="PLUS_TIMES",
         expect=[[1], [2]]),
    dict(cite="tests/test_vector.py:310 (m.transpose(), desc=T1)", op="vxm", A=[[1, 2, 0, 3], [0, 1, 2, 0], [1, 2, 3, 4], 4, 3], u=v3, type="INT64", semiring="PLUS_TIMES", desc="T1",
         expect=[[0, 1, 2, 3], [12, 2, 6, 8]]),
    dict(cite="tests/test_vector.py:312-315 (INT64.PLUS_PLUS)", op="vxm", A=[[0, 1, 2, 0], [1, 2, 0, 3], [1, 2, 3, 4], 3, 4], u=v3, type="INT64", semiring="PLUS_PLUS", expect=[[0, 1, 2, 3], [7, 3, 5, 6]]),
    # ---- doctests pygraphblas/vector.py:854-938 (vxm)
    dict(cite="pygraphblas/vector.py:858-863", op="vxm", A=m3, u=v3, type="INT64", semiring="PLUS_TIMES", expect=[[0, 1, 2], [12, 2, 6]]),
    dict(cite="pygraphblas/vector.py:878-884 (accum=INT64.plus, out=o=v.dup())", op="vxm", A=m3, u=v3, w=v3, type="INT64", semiring="PLUS_TIMES", accum="PLUS", expect=[[0, 1, 2], [14, 5, 10]]),
    dict(cite="pygraphblas/vector.py:885-891 (Accum(INT64.min), o @= M)", op="vxm", A=m3, u=v3, w=v3, type="INT64", semiring="PLUS_TIMES", accum="MIN", expect=[[0, 1, 2], [2, 2, 4]]),
    dict(cite="pygraphblas/vector.py:897-901 (INT64.min_plus)", op="vxm", A=m3, u=v3, type="INT64", semiring="MIN_PLUS", expect=[[0, 1, 2], [7, 3, 5]]),
    dict(cite="pygraphblas/vector.py:932-937 (mask = o with o[1] deleted)", op="vxm", A=m3, u=v3, mask=[[0, 2], [12, 6], 3], type="INT64", semiring="PLUS_TIMES", expect=[[0, 2], [12, 6]]),
    # ---- tests/test_descriptor.py:13-30 (BOOL, out aliases the operand, empty mask, complement + replace)
    dict(cite="tests/test_descriptor.py:13-20 test_RCT0", op="mxv", A=[[0, 1, 2], [1, 2, 0], [True, True, True], 3, 3], A_type="BOOL", u=[[0], [True], 3], u_type="BOOL", w=[[0], [True], 3], w_type="BOOL",
         mask=[[], [], 3], mask_type="BOOL", type="BOOL", semiring="LOR_LAND", desc="RCT0", expect=[[1], [True]], expect_type="BOOL"),
    dict(cite="tests/test_descriptor.py:23-30 test_RC", op="mxv", A=[[0, 1, 2], [1, 2, 0], [True, True, True], 3, 3], A_type="BOOL", u=[[0], [True], 3], u_type="BOOL", w=[[0], [True], 3], w_type="BOOL",
         mask=[[], [], 3], mask_type="BOOL", type="BOOL", semiring="LOR_LAND", desc="RC", expect=[[2], [True]], expect_type="BOOL"),
]

# result types of `m @ n` (tests/test_matrix.py:1017-1028 test_promotion)
promotion = [dict(cite="tests/test_matrix.py:1017-1028", left="FP32", right="FP64", result="FP64"),
             dict(cite="tests/test_matrix.py:1017-1028", left="FP32", right="UINT8", result="FP32"),
             dict(cite="tests/test_matrix.py:1017-1028", left="INT8", right="UINT8", result="INT8")]

golden_answers = dict(karate_triangles=dict(cite="demo/Triangle-Counting.ipynb:33,56", value=45))

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")
with open(out, "w") as f:
    json.dump(dict(source="transcribed from /root/reference (Graphegon/pygraphblas) test sources; see 'cite' of each case",
                   cases=cases, promotion=promotion, golden_answers=golden_answers), f, indent=1)
print("wrote", out, len(cases), "cases")
