"""Drop-in stand-in for the third-party `suitesparse_graphblas` CFFI package that pygraphblas imports
(`from suitesparse_graphblas import lib, ffi, initialize, is_initialized`, pygraphblas/__init__.py:248,
pygraphblas/base.py:7).  It binds the MI355X backend instead of SuiteSparse:GraphBLAS:

    ffi.cdef(<include/grb_mi355x.h>)        # the header is written to be cdef()-able verbatim
    lib = ffi.dlopen("libgrb_mi355x.so")    # ABI mode: no compile step

so the *unmodified* reference package runs on the HIP kernels:

    PYTHONPATH=<repo>/shim:/root/reference  /opt/conda/bin/python3.9 -c "import pygraphblas"

Needs a Python with cffi (the image's /opt/conda/bin/python3.9); the default python3.10 has no cffi and uses
the ctypes mirror `pygraphblas_amd` instead.  See INTEGRATION.md.
"""
import os
import re

import cffi

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_HERE))
_HEADER = os.environ.get("GRB_MI355X_HEADER", os.path.join(_REPO, "include", "grb_mi355x.h"))
_LIB = os.environ.get("GRB_MI355X_LIB", os.path.join(_REPO, "pygraphblas_amd", "libgrb_mi355x.so"))

ffi = cffi.FFI()
with open(_HEADER) as _f:
    _src = _f.read()
# cffi knows FILE and the stdint types; the header has no preprocessor logic beyond integer #defines
ffi.cdef(_src)
lib = ffi.dlopen(_LIB)
# Resolve every function of the header now.  In ABI mode cffi binds a name at its first use, under the FFI object's
# (non-reentrant) lock — the same lock ffi.new() holds while it parses a C type string for the first time.  pygraphblas
# frees its handles from __del__ (lib.GrB_Matrix_free, pygraphblas/matrix.py:117): a garbage collection that runs inside
# such a parse and is the first ever to need a *_free would wait for a lock its own thread holds.  (The compiled,
# API-mode suitesparse_graphblas binds everything at import, as this loop does.)
for _name in re.findall(r"^GrB_Info\s+(\w+)\s*\(", _src, flags=re.M):
    try:
        getattr(lib, _name)
    except AttributeError:      # declared, not exported: stays an AttributeError at the call site
        pass

__version__ = "5.1.0+mi355x"
_initialized = False


def is_initialized():
    return _initialized


def initialize(*, blocking=False, memory_manager="numpy"):
    """Same signature as suitesparse_graphblas.initialize (pygraphblas/__init__.py:251-256)."""
    global _initialized
    if _initialized:
        raise RuntimeError("GraphBLAS is already initialized!  Unable to initialize again.")
    info = lib.GrB_init(lib.GrB_BLOCKING if blocking else lib.GrB_NONBLOCKING)
    if info != lib.GrB_SUCCESS:
        raise RuntimeError(f"GrB_init failed with {info}")
    _initialized = True


def supports_complex():
    return False
