#!/usr/bin/env python3
"""Writes tests/golden/reference_companion_vectors.json — the golden vectors the reference's own tests hold for the companion
operations of the hot path (eWiseAdd / eWiseMult / apply / bound-scalar apply / select / transpose /
reduce to a scalar and to a vector / pattern / cast on matrices and vectors).

As for reference_vectors.json the reference cannot be executed here (its arithmetic is SuiteSparse:GraphBLAS, not installed, not under
/root/reference), so the vectors are TRANSCRIBED from the reference's test sources; every case cites the file:line it was read from, and
only values that stand literally in the reference's assertions are taken (no case is derived from a rule).
Format of a case:  kind: matrix | vector;  op: eadd | emult | apply | apply_first | apply_second | select | transpose | reduce | reduce_vector |
pattern | cast;  A / B: [I, J, X, nrows, ncols] (matrix) or [I, X, size] (vector);  type: the operands' type;  binop / unop / monoid / select /
thunk / scalar / to / desc as the op needs;  expect: [I, J, X] | [I, X] | a scalar  (+ expect_type when it differs from `type`).
"""
import json
import os

R10 = list(range(10))
diag10 = [R10, R10, R10, 10, 10]
cases = []


def M(cite, op, A, expect, type="INT64", **kw):
    cases.append(dict(cite=cite, kind="matrix", op=op, A=A, type=type, expect=expect, **kw))


def V(cite, op, A, expect, type="INT64", **kw):
    cases.append(dict(cite=cite, kind="vector", op=op, A=A, type=type, expect=expect, **kw))


# ---- tests/test_matrix.py:137-161 test_matrix_eadd:  v = diag(0..9) + (0,1)=1,  w = diag(0..9) + (1,0)=1
v_m = [R10 + [0], R10 + [1], R10 + [1], 10, 10]
w_m = [R10 + [1], R10 + [0], R10 + [1], 10, 10]
M("tests/test_matrix.py:137-151 (v.eadd(w) and v + w: addition_ref)", "eadd", v_m, [[0, 0] + R10[1:2] + [1] + R10[2:], [0, 1, 0, 1] + R10[2:], [0, 1, 1, 2] + list(range(4, 20, 2))], B=w_m, binop="PLUS")
# ---- tests/test_matrix.py:164-181 test_sub (the first of the two definitions): explicit zeros on the diagonal, 1 - empty = 1, empty - 1 copies the 1
M("tests/test_matrix.py:164-181 (v - w: subtraction_ref)", "eadd", v_m, [[0, 0, 1, 1] + R10[2:], [0, 1, 0, 1] + R10[2:], [0, 1, 1, 0] + [0] * 8], B=w_m, binop="MINUS")
# ---- tests/test_matrix.py:184-205 test_matrix_emult
V10 = list(range(1, 11))
dv = [R10, R10, V10, 10, 10]
M("tests/test_matrix.py:184-190 (v.emult(w))", "emult", dv, [R10, R10, [x * x for x in V10]], B=dv, binop="TIMES")
M("tests/test_matrix.py:199-201 (v / w: division_ref)", "emult", dv, [R10, R10, [1] * 10], B=dv, binop="DIV")
# ---- tests/test_matrix.py:208-240 reductions to a scalar
M("tests/test_matrix.py:208-210 (empty BOOL matrix)", "reduce", [[], [], [], 10, 10], False, type="BOOL", monoid="LOR", to="BOOL")
M("tests/test_matrix.py:211-213", "reduce", [[3, 4], [3, 4], [True, False], 10, 10], True, type="BOOL", monoid="LOR", to="BOOL")
M("tests/test_matrix.py:214-215 (with BOOL.LAND_MONOID)", "reduce", [[3, 4], [3, 4], [True, False], 10, 10], False, type="BOOL", monoid="LAND", to="BOOL")
M("tests/test_matrix.py:218-222 (empty INT8 matrix)", "reduce", [[], [], [], 10, 10], 0, type="INT8", monoid="PLUS", to="INT64")
M("tests/test_matrix.py:223-225", "reduce", [[3, 4], [3, 4], [3, 4], 10, 10], 7, type="INT8", monoid="PLUS", to="INT64")
M("tests/test_matrix.py:226-227 (with INT8.TIMES_MONOID)", "reduce", [[3, 4], [3, 4], [3, 4], 10, 10], 12, type="INT8", monoid="TIMES", to="INT64")
M("tests/test_matrix.py:230-234 (empty FP64 matrix)", "reduce", [[], [], [], 10, 10], 0.0, type="FP64", monoid="PLUS", to="FP64")
M("tests/test_matrix.py:235-237", "reduce", [[3, 4], [3, 4], [3.3, 4.4], 10, 10], 7.7, type="FP64", monoid="PLUS", to="FP64")
M("tests/test_matrix.py:238-240 (FP64.TIMES_MONOID)", "reduce", [[3, 4], [3, 4], [3.3, 4.4], 10, 10], 14.52, type="FP64", monoid="TIMES", to="FP64")
# ---- tests/test_matrix.py:243-246 test_matrix_reduce_vector
M("tests/test_matrix.py:243-246", "reduce_vector", diag10, [R10, R10], monoid="PLUS")
# ---- tests/test_matrix.py:309-315 test_matrix_pattern
M("tests/test_matrix.py:309-315 (BOOL, 10 x 10, nvals 10)", "pattern", diag10, [R10, R10, [True] * 10], expect_type="BOOL")
# ---- tests/test_matrix.py:318-326 test_matrix_transpose
vt = [[2, 1, 0], [0, 1, 2], [0, 1, 2], 3, 4]
M("tests/test_matrix.py:318-323", "transpose", vt, [[0, 1, 2], [2, 1, 0], [0, 1, 2]], expect_shape=[4, 3])
M("tests/test_matrix.py:324-325 (desc=T0: the matrix itself)", "transpose", vt, [[0, 1, 2], [2, 1, 0], [2, 1, 0]], desc="T0", expect_shape=[3, 4])
# ---- tests/test_matrix.py:536-545 test_apply
M("tests/test_matrix.py:536-539", "apply", [[0, 1, 2], [0, 1, 2], [2, 3, 4], 3, 3], [[0, 1, 2], [0, 1, 2], [-2, -3, -4]], unop="AINV")
# ---- tests/test_matrix.py:580-603 test_select
sv = [[0, 1, 2], [0, 1, 2], [0, 0, 3], 3, 3]
M("tests/test_matrix.py:580-583 (lib.GxB_NONZERO)", "select", sv, [[2], [2], [3]], select="NONZERO")
M("tests/test_matrix.py:585-586 ('!=0')", "select", sv, [[2], [2], [3]], select="!=0")
M("tests/test_matrix.py:588-589 ('!=', 0)", "select", sv, [[2], [2], [3]], select="!=", thunk=0)
M("tests/test_matrix.py:591-592 ('>', 0)", "select", sv, [[2], [2], [3]], select=">", thunk=0)
M("tests/test_matrix.py:594-595 ('<', 3)", "select", sv, [[0, 1], [0, 1], [0, 0]], select="<", thunk=3)
M("tests/test_matrix.py:597-598 ('>=', 0: the matrix itself)", "select", sv, [[0, 1, 2], [0, 1, 2], [0, 0, 3]], select=">=", thunk=0)
M("tests/test_matrix.py:600-601 ('>=0')", "select", sv, [[0, 1, 2], [0, 1, 2], [0, 0, 3]], select=">=0")
# ---- tests/test_matrix.py:608-658 test_select_ops:  m = the full 3 x 3 matrix of 0..8
full_I = [0, 0, 0, 1, 1, 1, 2, 2, 2]; full_J = [0, 1, 2, 0, 1, 2, 0, 1, 2]
m9 = [full_I, full_J, list(range(9)), 3, 3]
M("tests/test_matrix.py:613-615 (m.tril())", "select", m9, [[0, 1, 1, 2, 2, 2], [0, 0, 1, 0, 1, 2], [0, 3, 4, 6, 7, 8]], select="TRIL")
M("tests/test_matrix.py:617-619 (m.triu())", "select", m9, [[0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2], [0, 1, 2, 4, 5, 8]], select="TRIU")
M("tests/test_matrix.py:621 (m.diag())", "select", m9, [[0, 1, 2], [0, 1, 2], [0, 4, 8]], select="DIAG")
M("tests/test_matrix.py:623-625 (m.offdiag())", "select", m9, [[0, 0, 1, 1, 2, 2], [1, 2, 0, 2, 0, 1], [1, 2, 3, 5, 6, 7]], select="OFFDIAG")
M("tests/test_matrix.py:627-631 (m.nonzero())", "select", m9, [[0, 0, 1, 1, 1, 2, 2, 2], [1, 2, 0, 1, 2, 0, 1, 2], [1, 2, 3, 4, 5, 6, 7, 8]], select="NONZERO")
M("tests/test_matrix.py:633-639 (-m)", "apply", m9, [full_I, full_J, [0, -1, -2, -3, -4, -5, -6, -7, -8]], unop="AINV")
M("tests/test_matrix.py:643-649 (abs(m))", "apply", m9, [full_I, full_J, list(range(9))], unop="ABS")
M("tests/test_matrix.py:651-654 (~m on FP64: MINV)", "apply", [[0, 1, 2], [0, 1, 2], [0.0, 1.0, 2.0], 3, 3], [[0, 1, 2], [0, 1, 2], [float("inf"), 1.0, 0.5]], type="FP64", unop="MINV")
# ---- tests/test_matrix.py:909-914 bound scalars
M("tests/test_matrix.py:909-910 (apply_first(2, INT8.PLUS))", "apply_first", [[0, 1], [0, 1], [4, 2], 2, 2], [[0, 1], [0, 1], [6, 4]], binop="PLUS", scalar=2)
M("tests/test_matrix.py:913-914 (apply_second(INT8.MINUS, 2))", "apply_second", [[0, 1], [0, 1], [5, 1], 2, 2], [[0, 1], [0, 1], [3, -1]], binop="MINUS", scalar=2)
# ---- tests/test_matrix.py:917-1009 scalar and matrix arithmetic through the operators
m51 = [[0, 1], [0, 1], [5, 1], 2, 2]
M("tests/test_matrix.py:920 (m + 3)", "apply_second", m51, [[0, 1], [0, 1], [8, 4]], binop="PLUS", scalar=3)
M("tests/test_matrix.py:921 (m + n)", "eadd", m51, [[0, 1], [0, 1], [10, 2]], B=m51, binop="PLUS")
M("tests/test_matrix.py:922 (3 + m)", "apply_first", m51, [[0, 1], [0, 1], [8, 4]], binop="PLUS", scalar=3)
M("tests/test_matrix.py:939 (m - 3)", "apply_second", m51, [[0, 1], [0, 1], [2, -2]], binop="MINUS", scalar=3)
M("tests/test_matrix.py:940 (m - n)", "eadd", m51, [[0, 1], [0, 1], [0, 0]], B=m51, binop="MINUS")
M("tests/test_matrix.py:941 (3 - m)", "apply_first", m51, [[0, 1], [0, 1], [-2, 2]], binop="MINUS", scalar=3)
M("tests/test_matrix.py:956 (m * 3)", "apply_second", m51, [[0, 1], [0, 1], [15, 3]], binop="TIMES", scalar=3)
M("tests/test_matrix.py:958 (m * n)", "emult", m51, [[0, 1], [0, 1], [25, 1]], B=m51, binop="TIMES")
M("tests/test_matrix.py:963 (3 * m)", "apply_first", m51, [[0, 1], [0, 1], [15, 3]], binop="TIMES", scalar=3)
M("tests/test_matrix.py:977-978 (m / 3)", "apply_second", [[0, 1], [0, 1], [15, 3], 2, 2], [[0, 1], [0, 1], [5, 1]], binop="DIV", scalar=3)
M("tests/test_matrix.py:982-983 (15 / m)", "apply_first", [[0, 1], [0, 1], [3, 5], 2, 2], [[0, 1], [0, 1], [5, 3]], binop="DIV", scalar=15)
M("tests/test_matrix.py:990-991 (m /= n)", "emult", [[0, 1], [0, 1], [5, 1], 2, 2], [[0, 1], [0, 1], [1, 1]], B=[[0, 1], [0, 1], [5, 1], 2, 2], binop="DIV")
# ---- tests/test_matrix.py:1012-1015 test_cast
M("tests/test_matrix.py:1012-1015", "cast", [[0, 1], [0, 1], [4, 2], 2, 2], [[0, 1], [0, 1], [4.0, 2.0]], to="FP64", expect_type="FP64")

# ---- tests/test_vector.py:98-113 test_vector_eadd:  v = {0: 1, k: k for k = 2..9},  w = {1: 1, k: k}
K = list(range(2, 10))
vv = [[0] + K, [1] + K, 10]
wv = [[1] + K, [1] + K, 10]
V("tests/test_vector.py:98-109 (v.eadd(w): addition_ref)", "eadd", vv, [R10, [1, 1] + list(range(4, 20, 2))], B=wv, binop="PLUS")
V("tests/test_vector.py:119-122 (v - w: subtraction_ref)", "eadd", vv, [R10, [1, 1] + [0] * 8], B=wv, binop="MINUS")
V("tests/test_vector.py:127-131 (1 - v)", "apply_first", vv, [[0] + K, [0, -1, -2, -3, -4, -5, -6, -7, -8]], binop="MINUS", scalar=1)
V("tests/test_vector.py:133-135 (v - 1)", "apply_second", vv, [[0] + K, [0, 1, 2, 3, 4, 5, 6, 7, 8]], binop="MINUS", scalar=1)
V("tests/test_vector.py:137-139 (1 + v)", "apply_first", vv, [[0] + K, [2, 3, 4, 5, 6, 7, 8, 9, 10]], binop="PLUS", scalar=1)
V("tests/test_vector.py:141-143 (v + 1)", "apply_second", vv, [[0] + K, [2, 3, 4, 5, 6, 7, 8, 9, 10]], binop="PLUS", scalar=1)
V("tests/test_vector.py:157-161 (w = v.dup(); w += v)", "eadd", vv, [[0] + K, [2, 4, 6, 8, 10, 12, 14, 16, 18]], B=vv, binop="PLUS")
# ---- tests/test_vector.py:166-195 test_vector_emult
v110 = [R10, V10, 10]
V("tests/test_vector.py:166-171 (v.emult(w))", "emult", v110, [R10, [x * x for x in V10]], B=v110, binop="TIMES")
V("tests/test_vector.py:177-178 (v.emult(w, '+'))", "emult", v110, [R10, [x + x for x in V10]], B=v110, binop="PLUS")
V("tests/test_vector.py:180-183 (v / w: division_ref)", "emult", v110, [R10, [1] * 10], B=v110, binop="DIV")
# ---- tests/test_vector.py:198-213 test_vector_pattern
V("tests/test_vector.py:198-207 (v.pattern())", "pattern", [[0, 2], [0, 42], 3], [[0, 2], [True, True]], to="BOOL", expect_type="BOOL")
V("tests/test_vector.py:209-213 (v.pattern(INT8))", "pattern", [[0, 2], [0, 42], 3], [[0, 2], [1, 1]], to="INT8", expect_type="INT8")
# ---- tests/test_vector.py:216-240 reductions to a scalar
V("tests/test_vector.py:216-218 (empty BOOL vector)", "reduce", [[], [], 10], False, type="BOOL", monoid="LOR", to="BOOL")
V("tests/test_vector.py:219-220", "reduce", [[3], [True], 10], True, type="BOOL", monoid="LOR", to="BOOL")
V("tests/test_vector.py:223-227 (empty INT64 vector)", "reduce", [[], [], 10], 0, monoid="PLUS", to="INT64")
V("tests/test_vector.py:228-230", "reduce", [[3, 4], [3, 4], 10], 7, monoid="PLUS", to="INT64")
V("tests/test_vector.py:233-237 (empty FP64 vector)", "reduce", [[], [], 10], 0.0, type="FP64", monoid="PLUS", to="FP64")
V("tests/test_vector.py:238-240", "reduce", [[3, 4], [3.3, 4.4], 10], 7.7, type="FP64", monoid="PLUS", to="FP64")
# ---- tests/test_vector.py:318-328 test_apply (an FP64 vector; ~v is MINV)
V("tests/test_vector.py:318-322 (v.apply(INT64.AINV) on FP64 values)", "apply", [[0, 1, 2], [2.0, 4.0, 8.0], 3], [[0, 1, 2], [-2.0, -4.0, -8.0]], type="FP64", unop="AINV", unop_type="INT64")
V("tests/test_vector.py:327-328 (~v)", "apply", [[0, 1, 2], [2.0, 4.0, 8.0], 3], [[0, 1, 2], [0.5, 0.25, 0.125]], type="FP64", unop="MINV")
# ---- tests/test_vector.py:331-334 test_select
V("tests/test_vector.py:331-334 (lib.GxB_NONZERO)", "select", [[0, 1, 2], [0, 0, 3], 3], [[2], [3]], select="NONZERO")
# ---- tests/test_vector.py:439-515 bound scalars and scalar arithmetic
V("tests/test_vector.py:439-441 (apply_first(2, INT8.PLUS))", "apply_first", [[0, 1], [4, 2], 2], [[0, 1], [6, 4]], binop="PLUS", scalar=2)
V("tests/test_vector.py:445-447 (apply_second(INT8.MINUS, 2))", "apply_second", [[0, 1], [5, 1], 2], [[0, 1], [3, -1]], binop="MINUS", scalar=2)
v51 = [[0, 1], [5, 1], 2]
V("tests/test_vector.py:454-456 (m + 3)", "apply_second", v51, [[0, 1], [8, 4]], binop="PLUS", scalar=3)
V("tests/test_vector.py:459-461 (3 + m)", "apply_first", v51, [[0, 1], [8, 4]], binop="PLUS", scalar=3)
V("tests/test_vector.py:470-472 (m - 3)", "apply_second", v51, [[0, 1], [2, -2]], binop="MINUS", scalar=3)
V("tests/test_vector.py:475-477 (3 - m)", "apply_first", v51, [[0, 1], [-2, 2]], binop="MINUS", scalar=3)
V("tests/test_vector.py:486-488 (m * 3)", "apply_second", v51, [[0, 1], [15, 3]], binop="TIMES", scalar=3)
V("tests/test_vector.py:491-493 (3 * m)", "apply_first", v51, [[0, 1], [15, 3]], binop="TIMES", scalar=3)
V("tests/test_vector.py:502-504 (m / 3)", "apply_second", [[0, 1], [15, 3], 2], [[0, 1], [5, 1]], binop="DIV", scalar=3)
V("tests/test_vector.py:507-509 (15 / m)", "apply_first", [[0, 1], [3, 5], 2], [[0, 1], [5, 3]], binop="DIV", scalar=15)
# ---- tests/test_vector.py:548-560 test_nonzero, test_neg_abs
V("tests/test_vector.py:548-550 (m.nonzero())", "select", [[0, 1], [0, 2], 2], [[1], [2]], select="NONZERO")
V("tests/test_vector.py:553-555 (-m)", "apply", [[0, 1], [0, 2], 2], [[0, 1], [0, -2]], unop="AINV")
V("tests/test_vector.py:557-558 (abs(m))", "apply", [[0, 1], [0, -2], 2], [[0, 1], [0, 2]], unop="ABS")

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_companion_vectors.json")
with open(out, "w") as f:
    json.dump(dict(source="transcribed from /root/reference (Graphegon/pygraphblas) tests/test_matrix.py and tests/test_vector.py; see 'cite' of each case", cases=cases), f, indent=1)
print("wrote", out, len(cases), "cases")
