"""The `suitesparse_graphblas`-compatible CFFI shim (shim/): the drop-in boundary the unmodified reference binds.

Python 3.10 here has no cffi, so these tests drive the image's /opt/conda/bin/python3.9 in a subprocess and skip
when it is missing.  With /root/reference present (build container only) the *unmodified* reference package is
imported on top of the shim and its own GPU-free tests are run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY39 = "/opt/conda/bin/python3.9"
REF = "/root/reference"


def have_py39():
    if not os.path.exists(PY39):
        return False
    return subprocess.run([PY39, "-c", "import cffi"], capture_output=True).returncode == 0


needs39 = pytest.mark.skipif(not have_py39(), reason="no /opt/conda/bin/python3.9 with cffi")


@needs39
def test_cffi_boundary_without_gpu(gb):
    if gb.device_info()["ok"]:
        pytest.skip("a HIP device is present")
    r = subprocess.run([PY39, os.path.join(ROOT, "tests", "shim_cffi_smoke.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK cpu" in r.stdout, r.stdout + r.stderr


@needs39
@pytest.mark.gpu
def test_cffi_boundary_on_gpu(gpu):
    r = subprocess.run([PY39, os.path.join(ROOT, "tests", "shim_cffi_smoke.py"), "--gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK gpu" in r.stdout, r.stdout + r.stderr


@needs39
@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_unmodified_reference_imports_and_runs_its_gpu_free_tests(gb):
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "shim") + ":" + REF, PYTHONDONTWRITEBYTECODE="1")   # the reference tree is read-only to us: no __pycache__ there
    code = ("import pygraphblas as p; from pygraphblas import *; "
            "assert INT64.PLUS_TIMES.ztype is INT64 and BOOL.LOR_LAND.ztype is BOOL and FP32.PLUS_SECOND is not None; "
            "m = Matrix.from_lists([0,1,2],[1,2,0],[1,2,3]); assert m.nvals == 3 and m[0,1] == 1 and m.type is INT64; "
            "v = Vector.from_lists([0,1,2],[2,3,4]); assert list(v) == [(0,2),(1,3),(2,4)]; "
            "assert descriptor.T1 in descriptor.CT1; print('OK import', len([n for n in dir(p.lib) if 'SEMIRING' in n or n.startswith('GxB_PLUS_')]))")
    r = subprocess.run([PY39, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd="/tmp")
    assert r.returncode == 0 and "OK import" in r.stdout, r.stdout + r.stderr
    # the reference's own tests that need no arithmetic pass unmodified
    # selected by name (-k): everything in these files that involves no arithmetic
    names = ("test_type_lookup_name test_gb_from_type test_promotion test_options_set test_descriptor test_scalar_create_from_type "
             "test_scalar_from_value test_scalar_dup test_scalar_clear test_scalar_wait test_matrix_init_without_type "
             "test_matrix_get_set_element test_clear test_resize test_matrix_create_dup test_matrix_to_from_lists test_matrix_gb_type "
             "test_matrix_random test_iters test_identity test_delitem test_vector_init_without_type test_vector_create_sparse "
             "test_vector_gb_type test_vector_create_dup test_vector_from_list test_vector_to_lists test_contains").split()
    files = [f"{REF}/tests/{f}" for f in ("test_types.py", "test_base.py", "test_descriptor.py", "test_scalar.py", "test_matrix.py", "test_vector.py")]
    r = subprocess.run([PY39, "-m", "pytest", "-c", "/dev/null", "--rootdir", "/tmp", "-p", "no:cacheprovider", "-q", "-k", " or ".join(names), *files],
                       capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr
    import re
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 28, r.stdout[-3000:]
    # -k matches substrings, so a few arithmetic tests ride along: without a GPU they must fail loudly (Panic), nothing else
    other = [l for l in r.stdout.splitlines() if l.startswith("FAILED") and "Panic" not in l]
    assert not other, other


@needs39
@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_unmodified_reference_reads_its_own_grb_fixture_through_the_shim(gb, tmp_path):
    """`Matrix.from_binfile` of the reference (pygraphblas/matrix.py:489-497 -> suitesparse_graphblas.io.binary.binread, here
    shim/suitesparse_graphblas/io/binary.py) on docs/test_binfile.grb == docs/test_mm.mm; to_binfile round-trips."""
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "shim") + ":" + REF, PYTHONDONTWRITEBYTECODE="1")   # the reference tree is read-only to us: no __pycache__ there
    code = ("from pygraphblas import *; "
            f"M = Matrix.from_binfile('{REF}/docs/test_binfile.grb'); "
            "assert M.type is INT64 and M.shape == (7, 7) and M.nvals == 12; "
            "mm = [l.split() for l in open('" + REF + "/docs/test_mm.mm') if not l.startswith('%')][1:]; "
            "want = sorted((int(a) - 1, int(b) - 1, int(c)) for a, b, c in mm); "
            "I, J, X = M.to_lists(); assert sorted(zip(I, J, X)) == want; "
            f"M.to_binfile('{tmp_path}/rt.grb'); M2 = Matrix.from_binfile('{tmp_path}/rt.grb'); assert M2.to_lists() == M.to_lists(); print('OK grb')")
    r = subprocess.run([PY39, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd="/tmp")
    assert r.returncode == 0 and "OK grb" in r.stdout, r.stdout + r.stderr


@needs39
@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_unmodified_reference_stores_complex_entries_through_the_shim(gb):
    """Complex containers (FC32 / FC64) are host-side storage here: what the reference's tests ask of them
    (tests/test_matrix.py:56-60, 853-855; tests/test_vector.py:385-387) works without arithmetic."""
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "shim") + ":" + REF, PYTHONDONTWRITEBYTECODE="1")   # the reference tree is read-only to us: no __pycache__ there
    code = ("from pygraphblas import *\n"
            "m = Matrix.from_lists([0], [0], [0j]); assert m.type is FC64 and m.shape == (1, 1) and m.nvals == 1 and m[0, 0] == 0j\n"
            "m[0, 0] = 3 + 4j; assert m[0, 0] == 3 + 4j and m.to_lists() == [[0], [0], [3 + 4j]]\n"
            "v = Vector.sparse(FC32, 10); v[3] = 1.5 - 2j; assert v[3] == 1.5 - 2j and v.nvals == 1 and v.to_lists() == [[3], [1.5 - 2j]]\n"
            "d = Matrix.dense(FC64, 10, 10); assert d.nvals == 100 and d[9, 9] == 0j\n"
            "w = Vector.dense(FC64, 5, fill=2j); assert w.nvals == 5 and w[4] == 2j\n"
            "import pytest\n"
            "with pytest.raises(TypeError): d.to_arrays()\n"
            "r = Matrix.sparse(FP64, 2, 2); r[0, 1] = 2.5; c = Matrix.sparse(FC64, 2, 2); c[0, 1] = r[0, 1]; assert c[0, 1] == 2.5 + 0j\n"
            "print('OK complex')\n")
    r = subprocess.run([PY39, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd="/tmp")
    assert r.returncode == 0 and "OK complex" in r.stdout, r.stdout + r.stderr


@needs39
@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_scalar_assign_over_a_slice_of_a_hypersparse_or_complex_container(gb):
    """the position count of a scalar assign was taken from the raw `ni`, which for a slice is the
    GxB_RANGE / GxB_STRIDE sentinel (~2^63) — `H[0:3, 0:3] = 5` on a default-dimension matrix was refused as "more than 2^24
    positions".  (The reference's slices are inclusive: 0:2 names three positions.)"""
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "shim") + ":" + REF, PYTHONDONTWRITEBYTECODE="1")
    code = ("from pygraphblas import *\n"
            "H = Matrix.sparse(INT64); H[0:2, 0:2] = 5; assert H.nvals == 9 and H[2, 2] == 5\n"
            "H[0:4:2, 1] = 7; assert H[0, 1] == 7 and H[2, 1] == 7 and H[4, 1] == 7 and H[1, 1] == 5\n"
            "v = Vector.sparse(INT64); v[3:5] = 1; assert v.to_lists() == [[3, 4, 5], [1, 1, 1]]\n"
            "Z = Matrix.sparse(FC64, 10, 10); Z[1:3, 1:3] = 1j; assert Z.nvals == 9 and Z[2, 2] == 1j\n"
            "print('OK slices')\n")
    r = subprocess.run([PY39, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd="/tmp")
    assert r.returncode == 0 and "OK slices" in r.stdout, r.stdout + r.stderr


@needs39
@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_host_mirror_slices_against_a_model_through_the_unmodified_reference(gb):
    """`M[i]`, `M[:, j]`, `M[i] = v`, `M[a:b, c:d]`, `M[i, j] = x` / `del M[i, j]` work on the sorted tuples of the host mirror
    (grb_host_ops.cpp); 300 random cases against a dict model (tests/shim_host_slices_check.py)."""
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "shim") + ":" + REF, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([PY39, os.path.join(ROOT, "tests", "shim_host_slices_check.py")], capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
    assert r.returncode == 0 and "OK host slices" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@needs39
@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_host_mirror_index_operations_fuzzed_against_a_model(gb):
    """extract / assign (matrix, row, column, vector) with index lists, ranges, masks (valued, structural, complemented),
    accumulators and replace, through the unmodified reference, against a Python model of the GraphBLAS rules
    (tools/fuzz_host_ops.py; 12 000 cases were run while it was written)."""
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "shim") + ":" + REF, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([PY39, os.path.join(ROOT, "tools", "fuzz_host_ops.py"), "--cases", "600", "--seed", "11"], capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
    assert r.returncode == 0 and "fuzz host ops ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
