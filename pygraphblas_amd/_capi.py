"""ctypes binding of libgrb_mi355x.so — the C-ABI boundary of the MI355X GraphBLAS backend.

The reference binds the same entry points through CFFI (`from suitesparse_graphblas import lib, ffi`,
pygraphblas/__init__.py:248); Python 3.10 in this image has no cffi, so the host layer binds with
ctypes.  `names` enumerates the built-in handles by parsing include/grb_mi355x.h — the counterpart of
the reference's regex reflection over dir(lib) (pygraphblas/semiring.py:123-129).

There is NO CPU fallback: if the shared library is missing the import fails, and if no HIP device is
present every compute entry point returns GrB_PANIC (raised as `Panic`).
"""
import ctypes as C
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRB_MI355X_LIB", os.path.join(_PKG, "libgrb_mi355x.so"))
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "grb_mi355x.h")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950).  pygraphblas_amd has no CPU fallback.")

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

_decl_re = re.compile(r"^extern\s+(GrB_Type|GrB_UnaryOp|GrB_BinaryOp|GrB_Monoid|GrB_Semiring|GrB_Descriptor|GxB_SelectOp)\s+(\w+);")
_fn_re = re.compile(r"^GrB_Info\s+(\w+)\s*\(")
_def_re = re.compile(r"^#define\s+(\w+)\s+(-?\d+)\s*$")

names = {"GrB_Type": [], "GrB_UnaryOp": [], "GrB_BinaryOp": [], "GrB_Monoid": [], "GrB_Semiring": [],
         "GrB_Descriptor": [], "GxB_SelectOp": []}
functions = []
constants = {}
with open(HEADER_PATH) as _f:
    for _line in _f:
        _m = _decl_re.match(_line)
        if _m:
            names[_m.group(1)].append(_m.group(2))
            continue
        _m = _fn_re.match(_line)
        if _m:
            functions.append(_m.group(1))
            continue
        _m = _def_re.match(_line)
        if _m:
            constants[_m.group(1)] = int(_m.group(2))


_handles = {}


def handle(name):
    """Value of an exported handle variable (`extern GrB_Semiring NAME;`) — looked up in the library once (the symbol lookup is ~1 us, and
    `GrB_ALL` is asked for by every whole-vector assign of a loop)."""
    v = _handles.get(name)
    if v is None:
        v = _handles[name] = C.c_void_p.in_dll(lib, name).value
    return v


_all_ptr = None


def all_indices():
    """`GrB_ALL` as the `const GrB_Index *` argument of the assign / extract entry points (one object, made once)."""
    global _all_ptr
    if _all_ptr is None:
        _all_ptr = C.cast(handle("GrB_ALL"), C.c_void_p)
    return _all_ptr


# every GrB_* function returns GrB_Info (int)
missing = [_n for _n in functions if not hasattr(lib, _n)]   # must be empty (tests/test_capi_symbols.py)
for _n in functions:
    if _n not in missing:
        getattr(lib, _n).restype = C.c_int

vp = C.c_void_p
u64 = C.c_uint64
NULL = None


def init():
    info = lib.GrB_init(C.c_int(constants["GrB_NONBLOCKING"]))
    if info != 0:
        raise RuntimeError(f"GrB_init failed with {info}")


def device_info():
    buf = C.create_string_buffer(256)
    cus = C.c_int(0)
    hbm = C.c_size_t(0)
    info = lib.GrBX_device_info(buf, C.c_int(256), C.byref(cus), C.byref(hbm))
    return {"ok": info == 0, "name": buf.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}


def last_kernel_plan():
    buf = C.create_string_buffer(512)
    lib.GrBX_last_kernel_plan(buf, C.c_int(512))
    return buf.value.decode().strip()
