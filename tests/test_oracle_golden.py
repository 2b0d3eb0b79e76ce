"""Pins the CPU oracle to every golden vector the reference's tests and doctests hold for the mxm / mxv / vxm
path (tests/golden/reference_vectors.json; transcription script tests/golden/make_reference_vectors.py)."""
import pytest

from golden_runner import load, run_oracle

DATA = load()


@pytest.mark.parametrize("case", DATA["cases"], ids=[c["cite"] for c in DATA["cases"]])
def test_oracle_matches_reference_vector(case):
    got = run_oracle(case)
    assert got == case["expect"], f"{case['cite']}: oracle gives {got}, reference pins {case['expect']}"


def test_fixture_is_current():
    """The committed JSON is what the committed script writes."""
    import importlib.util, json, os, tempfile, shutil
    here = os.path.dirname(os.path.abspath(__file__))
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(os.path.join(here, "golden", "make_reference_vectors.py"), tmp)
        spec = importlib.util.spec_from_file_location("mk", os.path.join(tmp, "make_reference_vectors.py"))
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
        assert json.load(open(os.path.join(tmp, "reference_vectors.json"))) == DATA
    finally:
        shutil.rmtree(tmp)
