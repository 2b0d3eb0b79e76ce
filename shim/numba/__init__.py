"""Import-time stand-in for numba (test harness only).

pygraphblas imports numba unconditionally (pygraphblas/base.py:8, types.py:16-19, binaryop.py:19, unaryop.py:11-13,
selectop.py:24-25, matrix.py:40) but only *uses* it for user-defined operators, which the MI355X backend does not
support (host callbacks cannot run inside HIP kernels).  The image's numba build does not import
(numpy ABI mismatch), so this stub provides the names needed for `import pygraphblas` to succeed."""


class _Sig:
    def __init__(self, name):
        self.name = name

    def __call__(self, *a, **k):
        return self

    def __getitem__(self, item):
        return self

    def __repr__(self):
        return self.name


def _unsupported(*a, **k):
    def deco(fn):
        def call(*aa, **kk):
            raise NotImplementedError("user-defined operators need numba, which this environment does not provide")
        return call
    if len(a) == 1 and callable(a[0]) and not k:
        return deco(a[0])
    return deco


njit = jit = cfunc = _unsupported


def carray(*a, **k):
    raise NotImplementedError("numba.carray is not available")


void = _Sig("void")
boolean = _Sig("boolean")
int8 = _Sig("int8"); int16 = _Sig("int16"); int32 = _Sig("int32"); int64 = _Sig("int64")
uint8 = _Sig("uint8"); uint16 = _Sig("uint16"); uint32 = _Sig("uint32"); uint64 = _Sig("uint64")
float32 = _Sig("float32"); float64 = _Sig("float64"); complex64 = _Sig("complex64"); complex128 = _Sig("complex128")


class types:
    @staticmethod
    def CPointer(t):
        return _Sig(f"CPointer({t})")

    class Record:
        @staticmethod
        def make_c_struct(*a, **k):
            return _Sig("Record")
