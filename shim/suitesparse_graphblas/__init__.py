"""Drop-in stand-in for the third-party `suitesparse_graphblas` CFFI package that pygraphblas imports
(`from suitesparse_graphblas import lib, ffi, initialize, is_initialized`, pygraphblas/__init__.py:248,
pygraphblas/base.py:7).  It binds the MI355X backend instead of SuiteSparse:GraphBLAS:

    ffi.cdef(<include/grb_mi355x.h>)        # the header is written to be cdef()-able verbatim
    lib = ffi.dlopen("libgrb_mi355x.so")    # ABI mode: no compile step

so the *unmodified* reference package runs on the HIP kernels:

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=<repo>/shim:/root/reference  /opt/conda/bin/python3.9 -c "import pygraphblas"

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



class _Lib:
    """`lib` as pygraphblas sees it: the dlopen()ed library, with the complex-typed entry points made callable.

    The compiled (API-mode) suitesparse_graphblas passes `double _Complex` by value; cffi's ABI mode cannot (libffi
    call descriptions for complex arguments are refused), so the header declares GxB_FC32_t / GxB_FC64_t as
    {re, im} structs — the same register classes in the x86-64 SysV ABI — and the few *_FC32 / *_FC64 functions get a
    wrapper that turns a Python complex into that struct and a `double _Complex *` buffer (what pygraphblas/types.py:332,343
    allocates) into the struct pointer.  Every other name resolves straight to the library object."""

    def __init__(self, real, names):
        object.__setattr__(self, "_real", real)
        for name in names:
            try:
                fn = getattr(real, name)
            except AttributeError:
                continue
            m = re.search(r"_(FC32|FC64)$", name)
            self.__dict__[name] = self._complex_call(fn, "GxB_%s_t" % m.group(1)) if m else fn

    @staticmethod
    def _complex_call(fn, cname):
        kinds = [a.cname for a in ffi.typeof(fn).args]
        ptr = cname + " *"

        def call(*args):
            conv = []
            for a, k in zip(args, kinds):
                if k == cname:
                    a = complex(a)
                    a = (a.real, a.imag)
                elif k == ptr and isinstance(a, ffi.CData):
                    a = ffi.cast(ptr, a)
                conv.append(a)
            return fn(*conv)

        call.__name__ = getattr(fn, "__name__", "complex_call")
        return call

    def __getattr__(self, name):          # constants and handles: read through, at the time of the access
        return getattr(self._real, name)

    def __setattr__(self, name, value):
        setattr(self._real, name, value)

    def __dir__(self):
        return dir(self._real)


lib = _Lib(lib, re.findall(r"^GrB_Info\s+(\w+)\s*\(", _src, flags=re.M))

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
    """Complex containers can be built, set, read and listed (host side); arithmetic on them reports DomainMismatch."""
    return True
