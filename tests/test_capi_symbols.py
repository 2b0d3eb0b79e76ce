"""The C-ABI shared library loads without a GPU and exports every symbol include/grb_mi355x.h declares
(no compute calls here).  Also pins the error convention (positive GrB_Info, pygraphblas/base.py:189-203)
and that compute entry points fail loudly — never fall back to a CPU path — when no HIP device exists."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "grb_mi355x.h")


def declared():
    fns, handles, defs = [], [], {}
    for line in open(HEADER):
        m = re.match(r"^GrB_Info\s+(\w+)\s*\(", line)
        if m:
            fns.append(m.group(1))
        m = re.match(r"^extern\s+(?:const\s+)?\w+\s*\*?\s*(\w+);", line)
        if m:
            handles.append(m.group(1))
        m = re.match(r"^#define\s+(\w+)\s+(-?\d+)\s*$", line)
        if m:
            defs[m.group(1)] = int(m.group(2))
    return fns, handles, defs


def test_every_declared_symbol_is_exported(gb):
    fns, handles, _ = declared()
    assert len(fns) > 250 and len(handles) > 1800
    lib = C.CDLL(os.path.join(ROOT, "pygraphblas_amd", "libgrb_mi355x.so"))
    missing = [n for n in fns + handles if not hasattr(lib, n)]
    assert not missing, missing[:20]
    assert gb._capi.missing == []


def test_hot_path_entry_points_present(gb):
    for n in ("GrB_mxm", "GrB_mxv", "GrB_vxm", "GrB_Matrix_reduce_INT64", "GrB_Vector_reduce_BOOL", "GrB_Vector_assign_UINT8"):
        assert hasattr(gb.lib, n)


def test_error_codes_are_the_positive_v13_numbering():
    _, _, d = declared()
    exp = dict(GrB_SUCCESS=0, GrB_NO_VALUE=1, GrB_UNINITIALIZED_OBJECT=2, GrB_INVALID_OBJECT=3, GrB_NULL_POINTER=4, GrB_INVALID_VALUE=5,
               GrB_INVALID_INDEX=6, GrB_DOMAIN_MISMATCH=7, GrB_DIMENSION_MISMATCH=8, GrB_OUTPUT_NOT_EMPTY=9, GrB_OUT_OF_MEMORY=10,
               GrB_INSUFFICIENT_SPACE=11, GrB_INDEX_OUT_OF_BOUNDS=12, GrB_PANIC=13)
    for k, v in exp.items():
        assert d[k] == v


def test_semiring_names_the_reference_reflects_over(gb):
    # the names pygraphblas/semiring.py:87-121 and types.py:148-200 look for
    for t in ("INT8", "INT16", "INT32", "INT64", "UINT8", "UINT16", "UINT32", "UINT64", "FP32", "FP64"):
        for n in (f"GrB_PLUS_TIMES_SEMIRING_{t}", f"GrB_MIN_PLUS_SEMIRING_{t}", f"GxB_PLUS_PAIR_{t}", f"GxB_PLUS_SECOND_{t}", f"GxB_PLUS_FIRST_{t}",
                  f"GxB_ANY_PAIR_{t}", f"GxB_PLUS_TIMES_{t}", f"GrB_PLUS_MONOID_{t}", f"GxB_MIN_{t}_MONOID", f"GrB_PLUS_{t}", f"GrB_EQ_{t}"):
            assert gb._capi.handle(n), n
    for n in ("GrB_LOR_LAND_SEMIRING_BOOL", "GxB_LOR_LAND_BOOL", "GxB_ANY_PAIR_BOOL", "GrB_LOR_MONOID_BOOL", "GrB_LAND_MONOID_BOOL", "GrB_LOR", "GrB_LAND"):
        assert gb._capi.handle(n), n
    assert len(gb._capi.names["GrB_Descriptor"]) == 31 and len(gb._capi.names["GxB_SelectOp"]) == 16


def test_compute_fails_loudly_without_a_device(gb):
    if gb.device_info()["ok"]:
        pytest.skip("a HIP device is present")
    m = gb.Matrix.from_lists([0, 1, 2], [1, 2, 0], [1, 2, 3])
    v = gb.Vector.from_lists([0, 1, 2], [2, 3, 4])
    for call in (lambda: m @ v, lambda: v @ m, lambda: m @ m, lambda: m.reduce_int(), lambda: v.reduce_bool()):
        with pytest.raises(gb.Panic, match="no device"):
            call()


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under pygraphblas_amd/ may reference it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pygraphblas_amd")):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".inc")):
                s = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|liboracle|grb_oracle", s):
                    bad.append(f)
    assert not bad, bad
