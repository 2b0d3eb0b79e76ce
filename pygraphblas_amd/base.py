"""Error convention of the C ABI -> Python exceptions.

Same names and the same positive GrB_Info numbering as the reference (pygraphblas/base.py:132-210):
1 NoValue (a KeyError), 2..13 the API / execution errors.  The message comes from
GrB_Matrix_error / GrB_Vector_error like the reference's `_check` (pygraphblas/matrix.py:43-51).
"""
import ctypes as C

from ._capi import lib


class GraphBLASException(Exception):
    pass


class NoValue(GraphBLASException, KeyError):
    pass


class UninitializedObject(GraphBLASException):
    pass


class InvalidObject(GraphBLASException):
    pass


class NullPointer(GraphBLASException):
    pass


class InvalidValue(GraphBLASException):
    pass


class InvalidIndex(GraphBLASException):
    pass


class DomainMismatch(GraphBLASException):
    pass


class DimensionMismatch(GraphBLASException):
    pass


class OutputNotEmpty(GraphBLASException):
    pass


class OutOfMemory(GraphBLASException):
    pass


class InsufficientSpace(GraphBLASException):
    pass


class IndexOutOfBound(GraphBLASException):
    pass


class Panic(GraphBLASException):
    pass


_error_codes = {
    1: NoValue, 2: UninitializedObject, 3: InvalidObject, 4: NullPointer, 5: InvalidValue, 6: InvalidIndex,
    7: DomainMismatch, 8: DimensionMismatch, 9: OutputNotEmpty, 10: OutOfMemory, 11: InsufficientSpace,
    12: IndexOutOfBound, 13: Panic,
}


def check(info, obj=None):
    """Raise the exception mapped to a non-zero GrB_Info, with the object's error string."""
    if info == 0:
        return
    msg = ""
    if obj is not None:
        s = C.c_char_p()
        fn = lib.GrB_Matrix_error if obj._kind == "matrix" else lib.GrB_Vector_error
        if fn(C.byref(s), obj._h) == 0 and s.value:
            msg = s.value.decode()
    raise _error_codes.get(info, GraphBLASException)(msg or f"GrB_Info {info}")
