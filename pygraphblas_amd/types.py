"""Built-in GraphBLAS types and the operator objects that hang off them.

Mirror of the reference's type registry (pygraphblas/types.py:87-342): each type class carries its
binary operators, monoids and semirings as attributes in upper and lower case
(`FP64.PLUS_TIMES`, `INT64.min_plus`, `BOOL.LOR_LAND`), discovered from the names the C ABI exports
exactly like pygraphblas/semiring.py:87-129, monoid.py:81-101, binaryop.py:104-125 do over dir(lib).
Type promotion follows pygraphblas/types.py:465-500.
"""
import contextvars
import ctypes as C
import re

import numpy as np

from . import _capi
from ._capi import lib, handle
from .base import check

current_semiring = contextvars.ContextVar("current_semiring")
current_monoid = contextvars.ContextVar("current_monoid")
current_binop = contextvars.ContextVar("current_binop")
current_accum = contextvars.ContextVar("current_accum")


class _Op:
    _ctxvar = None

    def __init__(self, kind, cname, name, typ):
        self.kind, self.cname, self.name, self.type = kind, cname, name, typ
        self._h = handle(cname)
        self._token = None

    def get_op(self):
        return self._h

    def __repr__(self):
        return f"<{self.kind} {self.type.__name__}.{self.name}>"

    def __enter__(self):
        self._token = self._ctxvar.set(self)
        return self

    def __exit__(self, *exc):
        self._ctxvar.reset(self._token)
        return False


class UnaryOp(_Op):
    def __init__(self, cname, name, typ):
        super().__init__("UnaryOp", cname, name, typ)


class BinaryOp(_Op):
    """A built-in binary operator; as a context manager it is the default eWise operator."""
    _ctxvar = current_binop

    def __init__(self, cname, name, typ):
        super().__init__("BinaryOp", cname, name, typ)

    def __call__(self, A, B, *args, **kwargs):
        return A.emult(B, self, *args, **kwargs)


class Accum:
    """`with Accum(INT64.min): ...` — default accumulator (reference: pygraphblas/binaryop.py:80-101)."""

    def __init__(self, binaryop):
        self.binaryop = binaryop

    def __enter__(self):
        self._token = current_accum.set(self.binaryop)
        return self

    def __exit__(self, *exc):
        current_accum.reset(self._token)
        return False


class Monoid(_Op):
    _ctxvar = current_monoid

    def __init__(self, cname, name, typ):
        super().__init__("Monoid", cname, name, typ)


class Semiring(_Op):
    """`FP64.PLUS_TIMES(A, B)` dispatches on the operand classes like pygraphblas/semiring.py:47-56."""
    _ctxvar = current_semiring

    def __init__(self, cname, name, typ):
        super().__init__("Semiring", cname, name, typ)

    @property
    def ztype(self):
        # same introspection chain as pygraphblas/types.py:442-461
        m, b, t = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib.GxB_Semiring_add(C.byref(m), C.c_void_p(self._h)))
        check(lib.GxB_Monoid_operator(C.byref(b), m))
        check(lib.GxB_BinaryOp_ztype(C.byref(t), b))
        return Type._by_handle[t.value]

    def __call__(self, A, B, *args, **kwargs):
        from .matrix import Matrix
        from .vector import Vector
        if isinstance(A, Vector):
            return A.vxm(B, self, *args, **kwargs)
        if isinstance(B, Vector):
            return A.mxv(B, self, *args, **kwargs)
        return A.mxm(B, self, *args, **kwargs)


class MetaType(type):
    def __repr__(cls):
        return cls.__name__


class Type(metaclass=MetaType):
    _by_handle = {}
    _by_name = {}
    _gb_name = None
    _c = None
    _np = None
    default_zero = 0
    default_one = 1

    @classmethod
    def _default_semiring(cls):
        return cls.PLUS_TIMES

    @classmethod
    def _default_addop(cls):
        return cls.PLUS

    @classmethod
    def _default_multop(cls):
        return cls.TIMES


def _mk(name, gb, ctype, nptype, zero=0, one=1):
    cls = MetaType(name, (Type,), {"_gb_name": gb, "_c": ctype, "_np": nptype, "default_zero": zero, "default_one": one})
    cls._h = handle(gb)
    Type._by_handle[cls._h] = cls
    Type._by_name[name] = cls
    return cls


BOOL = _mk("BOOL", "GrB_BOOL", C.c_bool, np.bool_, False, True)
INT8 = _mk("INT8", "GrB_INT8", C.c_int8, np.int8)
UINT8 = _mk("UINT8", "GrB_UINT8", C.c_uint8, np.uint8)
INT16 = _mk("INT16", "GrB_INT16", C.c_int16, np.int16)
UINT16 = _mk("UINT16", "GrB_UINT16", C.c_uint16, np.uint16)
INT32 = _mk("INT32", "GrB_INT32", C.c_int32, np.int32)
UINT32 = _mk("UINT32", "GrB_UINT32", C.c_uint32, np.uint32)
INT64 = _mk("INT64", "GrB_INT64", C.c_int64, np.int64)
UINT64 = _mk("UINT64", "GrB_UINT64", C.c_uint64, np.uint64)
FP32 = _mk("FP32", "GrB_FP32", C.c_float, np.float32, 0.0, 1.0)
FP64 = _mk("FP64", "GrB_FP64", C.c_double, np.float64, 0.0, 1.0)

BOOL._default_semiring = classmethod(lambda cls: cls.LOR_LAND)
BOOL._default_addop = classmethod(lambda cls: cls.LOR)
BOOL._default_multop = classmethod(lambda cls: cls.LAND)

ALL_TYPES = (BOOL, INT8, UINT8, INT16, UINT16, INT32, UINT32, INT64, UINT64, FP32, FP64)
_T = "BOOL|UINT8|UINT16|UINT32|UINT64|INT8|INT16|INT32|INT64|FP32|FP64"

_semiring_re = re.compile(rf"^(?:GxB|GrB)_([A-Z]+)_([A-Z0-9]+)_(?:SEMIRING_)?({_T})$")
_monoid_gxb_re = re.compile(rf"^GxB_([A-Z]+)_({_T})_MONOID$")
_monoid_grb_re = re.compile(rf"^GrB_([A-Z]+)_MONOID_({_T})$")
_binop_re = re.compile(rf"^(?:GxB|GrB)_([A-Z0-9]+)_({_T})$")
_unop_re = _binop_re


def _attach(cls, name, obj):
    setattr(cls, name, obj)
    setattr(cls, name.lower(), obj)


def _build():
    for cname in _capi.names["GrB_Semiring"]:
        m = _semiring_re.match(cname)
        if m:
            add, mul, t = m.groups()
            _attach(Type._by_name[t], f"{add}_{mul}", Semiring(cname, f"{add}_{mul}", Type._by_name[t]))
    for cname in _capi.names["GrB_Monoid"]:
        m = _monoid_gxb_re.match(cname) or _monoid_grb_re.match(cname)
        if m:
            op, t = m.groups()
            _attach(Type._by_name[t], f"{op}_MONOID", Monoid(cname, f"{op}_MONOID", Type._by_name[t]))
    for cname in _capi.names["GrB_BinaryOp"]:
        m = _binop_re.match(cname)
        if m:
            op, t = m.groups()
            _attach(Type._by_name[t], op, BinaryOp(cname, op, Type._by_name[t]))
    for op in ("LOR", "LAND", "LXOR", "LXNOR"):
        _attach(BOOL, op, BinaryOp("GrB_" + op, op, BOOL))
    for cname in _capi.names["GrB_UnaryOp"]:
        m = _unop_re.match(cname)
        if m:
            op, t = m.groups()
            _attach(Type._by_name[t], op, UnaryOp(cname, op, Type._by_name[t]))


_build()

_promotion_order = (FP64, FP32, INT64, UINT64, INT32, UINT32, INT16, UINT16, INT8, UINT8)


def promote(left, right):
    """Result type of an operation inferred from its operand types (pygraphblas/types.py:484-500)."""
    if left == right:
        return left
    if left == BOOL:
        return right
    if right == BOOL:
        return left
    for t in _promotion_order:
        if t in (left, right):
            return t
    raise TypeError(f"inconvertable types {left!r} and {right!r}")


def from_python_value(v):
    """Type inferred from a Python value (reference: Matrix.from_lists with typ=None, matrix.py:305-314)."""
    if isinstance(v, (bool, np.bool_)):
        return BOOL
    if isinstance(v, (int, np.integer)):
        return INT64
    if isinstance(v, (float, np.floating)):
        return FP64
    raise TypeError(f"cannot infer a GraphBLAS type from {type(v)}")


def type_of_handle(h):
    return Type._by_handle[h]
