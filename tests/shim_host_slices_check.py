"""Run under /opt/conda/bin/python3.9 with PYTHONPATH=<repo>/shim:<reference>: the host-mirror slicing operations of the
backend (GrB_Col_extract, GrB_Row/Col_assign, GrB_Matrix_extract, setElement / removeElement + reads), driven through the
UNMODIFIED reference package, against a dict model — 300 random cases.  No GPU needed: these edit the host mirror."""
import random
from pygraphblas import *
from pygraphblas import descriptor

random.seed(7)


def model(M):
    return {(i, j): x for i, j, x in M}


def vmodel(v):
    return {i: x for i, x in v}


n = 40
for trial in range(300):
    A = Matrix.sparse(INT64, n, n)
    for _ in range(random.randint(0, 200)):
        A[random.randrange(n), random.randrange(n)] = random.randint(-9, 9)
    a = model(A)
    i, j = random.randrange(n), random.randrange(n)
    assert vmodel(A.extract_row(i)) == {c: x for (r, c), x in a.items() if r == i}, ("row", trial)
    assert vmodel(A.extract_col(j)) == {r: x for (r, c), x in a.items() if c == j}, ("col", trial)
    a0, a1 = sorted(random.sample(range(n), 2)); b0, b1 = sorted(random.sample(range(n), 2))
    assert model(A[a0:a1, b0:b1]) == {(r - a0, c - b0): x for (r, c), x in a.items() if a0 <= r <= a1 and b0 <= c <= b1}, ("slice", trial)
    St = A.extract_matrix(slice(a0, a1), slice(b0, b1), desc=descriptor.T0)
    assert model(St) == {(c - a0, r - b0): x for (r, c), x in a.items() if a0 <= c <= a1 and b0 <= r <= b1}, ("slice of the transpose", trial)
    v = Vector.sparse(INT64, n)
    for _ in range(random.randint(0, 15)):
        v[random.randrange(n)] = random.randint(-9, 9)
    vm = vmodel(v)
    B = A.dup(); B[i] = v
    exp = {k: x for k, x in a.items() if k[0] != i}; exp.update({(i, c): x for c, x in vm.items()})
    assert model(B) == exp, ("row assign", trial)
    B = A.dup(); B[:, j] = v
    exp = {k: x for k, x in a.items() if k[1] != j}; exp.update({(r, j): x for r, x in vm.items()})
    assert model(B) == exp, ("column assign", trial)
    B = A.dup(); B.assign_row(i, v, accum=INT64.PLUS); exp = dict(a)
    for c, x in vm.items():
        exp[(i, c)] = exp[(i, c)] + x if (i, c) in exp else x
    assert model(B) == exp, ("row assign with accum", trial)
    B = A.dup(); B.assign_col(j, v, accum=INT64.MIN); exp = dict(a)
    for r, x in vm.items():
        exp[(r, j)] = min(exp[(r, j)], x) if (r, j) in exp else x
    assert model(B) == exp, ("column assign with accum", trial)
    B = A.dup(); exp = dict(a)
    for _ in range(random.randint(1, 6)):
        p, q = random.randrange(n), random.randrange(n)
        if random.random() < 0.3 and (p, q) in exp:
            del B[p, q]; del exp[(p, q)]
        else:
            B[p, q] = 5; exp[(p, q)] = 5
    assert model(B) == exp and B.nvals == len(exp), ("element edits", trial)
print("OK host slices")
