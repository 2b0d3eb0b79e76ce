"""Predefined descriptors (`descriptor.T0`, `descriptor.RC`, ...) and the default-descriptor context.

Mirror of pygraphblas/descriptor.py:148-182: `T0`/`T1` transpose the first/second input, `C`
complements the mask, `S` makes it structural, `R` replaces the output; `a & b` combines.
"""
import contextvars
import ctypes as _ct

from . import _capi
from ._capi import lib, handle
from .base import check

current_desc = contextvars.ContextVar("current_desc")

_FIELDS = ("GrB_OUTP", "GrB_MASK", "GrB_INP0", "GrB_INP1")


class Descriptor:
    def __init__(self, h, name, owned=False):
        self._h, self.name, self._owned = h, name, owned
        self._token = None

    def get_desc(self):
        return self._h

    def _get(self, field):
        v = _ct.c_int(0)
        check(lib.GxB_Desc_get(_ct.c_void_p(self._h), _ct.c_int(_capi.constants[field]), _ct.byref(v)))
        return v.value

    def __and__(self, other):
        d = _ct.c_void_p()
        check(lib.GrB_Descriptor_new(_ct.byref(d)))
        for f in _FIELDS:
            for src in (self, other):
                val = src._get(f)
                if f == "GrB_MASK":
                    for bit in (_capi.constants["GrB_COMP"], _capi.constants["GrB_STRUCTURE"]):
                        if val & bit:
                            check(lib.GrB_Descriptor_set(d, _ct.c_int(_capi.constants[f]), _ct.c_int(bit)))
                elif val:
                    check(lib.GrB_Descriptor_set(d, _ct.c_int(_capi.constants[f]), _ct.c_int(val)))
        return Descriptor(d.value, self.name + other.name, owned=True)

    def __contains__(self, other):
        return all((self._get(f) & other._get(f)) == other._get(f) if f == "GrB_MASK" else
                   (other._get(f) == 0 or self._get(f) == other._get(f)) for f in _FIELDS)

    def __eq__(self, other):
        return isinstance(other, Descriptor) and all(self._get(f) == other._get(f) for f in _FIELDS)

    def __hash__(self):
        return hash(tuple(self._get(f) for f in _FIELDS))

    def __repr__(self):
        return f"<Descriptor {self.name}>"

    def __enter__(self):
        self._token = current_desc.set(self)
        return self

    def __exit__(self, *exc):
        current_desc.reset(self._token)
        return False

    def __del__(self):
        if getattr(self, "_owned", False) and self._h:
            h = _ct.c_void_p(self._h)
            lib.GrB_Descriptor_free(_ct.byref(h))
            self._h = None


__all__ = ["Descriptor", "current_desc"]
for _n in _capi.names["GrB_Descriptor"]:
    _short = _n[len("GrB_DESC_"):]
    globals()[_short] = Descriptor(handle(_n), _short)
    __all__.append(_short)
