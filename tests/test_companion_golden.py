"""The companion operations of the hot path against the golden vectors the reference's own tests hold for them
(tests/golden/reference_companion_vectors.json: 82 cases transcribed with file:line from tests/test_matrix.py:137-246, :309-326, :536-658, :909-1015 and
tests/test_vector.py:98-240, :318-334, :439-560).
  -m "not gpu": the small dictionary model of tests/companion_model.py reproduces every vector (the model is then what the fuzzers' models are measured by);
  -m gpu:       the HIP library, through the C ABI and the mirror's methods of the same names, reproduces every vector."""
import numpy as np
import pytest

import companion_model as CM

CASES = CM.load()
IDS = [f"{k:02d}-{c['kind']}-{c['op']}" for k, c in enumerate(CASES)]


def test_there_are_enough_vectors_and_every_one_cites_its_source():
    assert len(CASES) >= 80
    assert all(c["cite"].startswith("tests/test_matrix.py:") or c["cite"].startswith("tests/test_vector.py:") for c in CASES)
    assert {c["op"] for c in CASES} >= {"eadd", "emult", "apply", "apply_first", "apply_second", "select", "transpose", "reduce", "reduce_vector", "pattern", "cast"}


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_model_reproduces_the_reference_vector(case):
    got, want = CM.run(case), CM.expected(case)
    assert got == want, (case["cite"], got, want)


def _build(gb, case, key):
    T = getattr(gb, case["type"]); o = case[key]
    if case["kind"] == "matrix":
        return gb.Matrix.from_lists(list(o[0]), list(o[1]), [CM.wrap(case["type"], x) for x in o[2]], o[3], o[4], T)
    return gb.Vector.from_lists(list(o[0]), [CM.wrap(case["type"], x) for x in o[1]], o[2], T)


def run_product(gb, case):
    from pygraphblas_amd import descriptor as D
    T = getattr(gb, case["type"]); A = _build(gb, case, "A"); op = case["op"]
    if op == "eadd":
        out = A.eadd(_build(gb, case, "B"), getattr(T, case["binop"]))
    elif op == "emult":
        out = A.emult(_build(gb, case, "B"), getattr(T, case["binop"]))
    elif op == "apply":
        out = A.apply(getattr(getattr(gb, case.get("unop_type", case["type"])), case["unop"]))
    elif op == "apply_first":
        out = A.apply_first(case["scalar"], getattr(T, case["binop"]))
    elif op == "apply_second":
        out = A.apply_second(getattr(T, case["binop"]), case["scalar"])
    elif op == "select":
        out = A.select(case["select"], case.get("thunk"))
    elif op == "transpose":
        out = A.transpose(desc=D.T0) if case.get("desc") == "T0" else A.transpose()
        assert [out.nrows, out.ncols] == case["expect_shape"]
    elif op == "pattern":
        out = A.pattern(getattr(gb, case.get("to", "BOOL")))
    elif op == "cast":
        out = A.cast(getattr(gb, case["to"]))
    elif op == "reduce_vector":
        out = A.reduce_vector(getattr(T, case["monoid"] + "_MONOID"))
    elif op == "reduce":
        mon = getattr(T, case["monoid"] + "_MONOID")
        return {"BOOL": A.reduce_bool, "INT64": A.reduce_int, "FP64": A.reduce_float}[case["to"]](mon)
    else:
        raise ValueError(op)
    et = case.get("expect_type", case["type"])
    assert out.type.__name__ == et, (out.type.__name__, et)
    if case["kind"] == "matrix" and op != "reduce_vector":
        I, J, X = out.to_arrays()
        return sorted(((int(i), int(j)), CM.wrap(et, x)) for i, j, x in zip(I.tolist(), J.tolist(), X.tolist()))
    I, X = out.to_arrays()
    return sorted((int(i), CM.wrap(et, x)) for i, x in zip(I.tolist(), X.tolist()))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_hip_library_reproduces_the_reference_vector(gb, gpu, case):
    got, want = run_product(gb, case), CM.expected(case)
    if case["op"] == "reduce":
        assert type(got) is type(want) and got == want, (case["cite"], got, want)
    else:
        assert got == want, (case["cite"], got, want)
