"""`Vector` — the host-side mirror of pygraphblas.Vector for the vxm / mxv hot path and the O(n)
operations the reference's BFS and PageRank loops wrap around it.

Same names and argument meaning as the reference class (pygraphblas/vector.py): `Vector.sparse/dense/
from_lists/from_list`, `size/nvals`, `vxm` (:835-971), `@`, `iseq`, `reduce_bool/int/float` (:1101-1202),
`assign_scalar` (:1494-1524), `eadd/emult` (:604-833), `apply`, element get/set/del.
"""
import ctypes as C
from functools import partial

import numpy as np

from . import _capi, types, descriptor as _d
from ._capi import lib, u64
from .base import check, NoValue
from .matrix import get_args, _p
from .types import current_semiring, current_binop, current_monoid


class Vector:
    _kind = "vector"

    def __init__(self, handle, typ=None):
        self._h = handle
        if typ is None:
            t = C.c_void_p()
            check(lib.GxB_Vector_type(C.byref(t), self._h))
            typ = types.type_of_handle(t.value)
        self.type = typ

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and lib is not None:
            lib.GrB_Vector_free(C.byref(h))

    # ---- construction -----------------------------------------------------------------------------------
    @classmethod
    def sparse(cls, typ, size=None):
        h = C.c_void_p()
        check(lib.GrB_Vector_new(C.byref(h), C.c_void_p(typ._h), u64(_capi.constants["GxB_INDEX_MAX"] if size is None else size)))
        return cls(h, typ)

    @classmethod
    def dense(cls, typ, size, fill=None):
        v = cls.sparse(typ, size)
        v.assign_scalar(typ.default_zero if fill is None else fill)
        return v

    @classmethod
    def from_lists(cls, I, V, size=None, typ=None):
        if typ is None:
            typ = types.from_python_value(V[0])
        if size is None:
            size = max(I) + 1
        return cls.from_arrays(np.asarray(I, np.uint64), np.asarray(V, typ._np), size, typ)

    @classmethod
    def from_list(cls, V, typ=None):
        return cls.from_lists(list(range(len(V))), V, len(V), typ)

    @classmethod
    def from_arrays(cls, I, V, size, typ, dup=None):
        v = cls.sparse(typ, size)
        I = np.ascontiguousarray(I, np.uint64); V = np.ascontiguousarray(V, typ._np)
        fn = getattr(lib, "GrB_Vector_build_" + typ.__name__)
        check(fn(v._h, _p(I), _p(V), u64(len(I)), C.c_void_p(dup.get_op()) if dup is not None else None), v)
        return v

    @classmethod
    def from_dense_array(cls, values, typ, present=None, device=False):
        """Import a full (or bitmap) vector in one call; `device=True`: `values`/`present` are (HBM address, n)."""
        h = C.c_void_p()
        if device:
            addr, n = values
            check(lib.GrBX_Vector_import_Bitmap(C.byref(h), C.c_void_p(typ._h), u64(n), C.c_void_p(addr),
                                                C.c_void_p(present) if present else None, C.c_int(1)))
        else:
            values = np.ascontiguousarray(values, typ._np)
            pr = np.ascontiguousarray(present, np.uint8) if present is not None else None
            check(lib.GrBX_Vector_import_Bitmap(C.byref(h), C.c_void_p(typ._h), u64(len(values)), _p(values),
                                                _p(pr) if pr is not None else None, C.c_int(0)))
        return cls(h, typ)

    def dup(self):
        h = C.c_void_p()
        check(lib.GrB_Vector_dup(C.byref(h), self._h), self)
        return Vector(h, self.type)

    # ---- properties -------------------------------------------------------------------------------------
    @property
    def size(self):
        n = u64()
        check(lib.GrB_Vector_size(C.byref(n), self._h), self)
        return n.value

    @property
    def shape(self):
        return (self.size,)

    @property
    def nvals(self):
        n = u64()
        check(lib.GrB_Vector_nvals(C.byref(n), self._h), self)
        return n.value

    def __len__(self):
        return self.nvals

    def clear(self):
        check(lib.GrB_Vector_clear(self._h), self)

    def wait(self):
        check(lib.GrB_Vector_wait(C.byref(self._h)), self)

    # ---- element access -----------------------------------------------------------------------------------
    def to_arrays(self):
        n = self.nvals
        I, X = np.zeros(n, np.uint64), np.zeros(n, self.type._np)
        nn = u64(n)
        check(getattr(lib, "GrB_Vector_extractTuples_" + self.type.__name__)(_p(I), _p(X), C.byref(nn), self._h), self)
        return I, X

    def to_lists(self):
        I, X = self.to_arrays()
        return [I.tolist(), X.tolist()]

    def to_dense_arrays(self):
        """(values, present) as dense numpy arrays of length size."""
        n = self.size
        x, p = np.zeros(n, self.type._np), np.zeros(n, np.uint8)
        check(lib.GrBX_Vector_export_Bitmap(self._h, _p(x), _p(p), C.c_int(0)), self)
        return x, p

    def device_view(self):
        """(values HBM address, present HBM address, nvals) — valid until the vector is next written."""
        v, p, n = C.c_void_p(), C.c_void_p(), u64()
        check(lib.GrBX_Vector_device_view(self._h, C.byref(v), C.byref(p), C.byref(n)), self)
        return v.value, p.value, n.value

    def __iter__(self):
        I, X = self.to_arrays()
        return iter(zip(I.tolist(), X.tolist()))

    def __getitem__(self, i):
        """`v[i]` (element) or `v[a:b]` / `v[[..]]` (sub-vector; a slice includes its stop, as in the reference: vector.py:1526-1547)."""
        if isinstance(i, (slice, list, tuple, np.ndarray, range)):
            return self.extract(i)
        out = self.type._c()
        check(getattr(lib, "GrB_Vector_extractElement_" + self.type.__name__)(C.byref(out), self._h, u64(i)), self)
        return out.value

    def extract(self, index, out=None, mask=None, accum=None, desc=None):
        """`out<mask> = accum(out, self(I))` (reference: vector.py:1549-1575); host-mirror operation of the library."""
        from .matrix import build_range
        I, ni, size, keep = build_range(index, self.size - 1)
        if out is None:
            out = Vector.sparse(self.type, self.size if size is None else size)
        mh, ah, dh = get_args(mask, accum, desc)
        check(lib.GrB_Vector_extract(out._h, mh, ah, self._h, I, u64(ni), dh), out)
        return out

    def assign(self, value, index=None, mask=None, accum=None, desc=None):
        """`self(I)<mask> = accum(self(I), value)` (reference: vector.py:1454-1492)."""
        from .matrix import build_range
        I, ni, size, keep = build_range(index, self.size - 1)
        mh, ah, dh = get_args(mask, accum, desc)
        check(lib.GrB_Vector_assign(self._h, mh, ah, value._h, I, u64(ni), dh), self)

    def get(self, i, default=None):
        try:
            return self[i]
        except NoValue:
            return default

    def __setitem__(self, index, value):
        if isinstance(value, Vector):
            return self.assign(value, index if not isinstance(index, Vector) else None, mask=index if isinstance(index, Vector) else None)
        if isinstance(index, Vector):                       # `v[q] = level`: scalar assign under the mask q (reference: vector.py:1428-1441)
            return self.assign_scalar(value, mask=index)
        if isinstance(index, slice):
            if index != slice(None):
                raise NotImplementedError("only v[:] = scalar is supported for slices")
            self.assign_scalar(value)
            return
        check(getattr(lib, "GrB_Vector_setElement_" + self.type.__name__)(self._h, self.type._c(value), u64(index)), self)

    def __delitem__(self, i):
        check(lib.GrB_Vector_removeElement(self._h, u64(i)), self)

    def __contains__(self, i):
        return self.get(i) is not None

    # ---- the hot path ---------------------------------------------------------------------------------------
    def vxm(self, other, semiring=None, cast=None, out=None, mask=None, accum=None, desc=None):
        """Vector-matrix multiply `w<mask> = accum(w, self (+).(x) other)`  (reference: vector.py:835-971)."""
        if semiring is None:
            semiring = current_semiring.get(None)
        if out is None:
            transposed = desc is not None and _d.T1 in desc
            if semiring is not None:
                typ = semiring.ztype
            else:
                typ = cast or types.promote(self.type, other.type)
            out = Vector.sparse(typ, other.nrows if transposed else other.ncols)
        if semiring is None:
            semiring = out.type._default_semiring()
        mh, ah, dh = get_args(mask, accum, desc)
        check(lib.GrB_vxm(out._h, mh, ah, C.c_void_p(semiring.get_op()), self._h, other._h, dh), out)
        return out

    def __matmul__(self, other):
        return self.vxm(other)

    def __imatmul__(self, other):
        return self.vxm(other, out=self)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        typ = self.__dict__.get("type")
        op = getattr(typ, name, None) if typ is not None else None
        if isinstance(op, (types.Semiring, types.BinaryOp)):
            return partial(op, self)
        raise AttributeError(name)

    # ---- reductions -------------------------------------------------------------------------------------------
    def _reduce_scalar(self, suffix, ctype, default_type, mon, accum, desc):
        if mon is None:
            mon = current_monoid.get(getattr(default_type, "LOR_MONOID" if default_type is types.BOOL else "PLUS_MONOID"))
        out = ctype(0)
        _, ah, dh = get_args(None, accum, desc)
        check(getattr(lib, "GrB_Vector_reduce_" + suffix)(C.byref(out), ah, C.c_void_p(mon.get_op()), self._h, dh), self)
        return out.value

    def reduce_bool(self, mon=None, accum=None, desc=None):
        """LOR-reduce to a Python bool (reference: vector.py:1132-1154); the BFS loop condition."""
        return self._reduce_scalar("BOOL", C.c_bool, types.BOOL, mon, accum, desc)

    def reduce_int(self, mon=None, accum=None, desc=None):
        return self._reduce_scalar("INT64", C.c_int64, types.INT64, mon, accum, desc)

    def reduce_float(self, mon=None, accum=None, desc=None):
        return self._reduce_scalar("FP64", C.c_double, types.FP64, mon, accum, desc)

    # ---- O(n) companions ------------------------------------------------------------------------------------------
    def assign_scalar(self, value, index=None, mask=None, accum=None, desc=None):
        """`w<mask>(:) = accum(w, value)` (reference: vector.py:1494-1524) — `v.assign_scalar(level, mask=q)` in BFS."""
        mh, ah, dh = get_args(mask, accum, desc)
        if index is None:
            I, ni = _capi.all_indices(), 0
            keep = None
        else:
            keep = np.ascontiguousarray([index] if np.isscalar(index) else index, np.uint64)
            I, ni = _p(keep), len(keep)
        fn = self.type.__dict__.get("_vector_assign_fn")
        if fn is None:
            fn = getattr(lib, "GrB_Vector_assign_" + self.type.__name__); setattr(self.type, "_vector_assign_fn", fn)      # (looked up once per type)
        check(fn(self._h, mh, ah, self.type._c(value), I, u64(ni), dh), self)

    def _ewise(self, stem, other, op, cast, out, mask, accum, desc, default):
        if op is None:
            op = current_binop.get(None) or default(types.promote(self.type, other.type))
        if out is None:
            out = Vector.sparse(cast or types.promote(self.type, other.type), self.size)
        mh, ah, dh = get_args(mask, accum, desc)
        fn = getattr(lib, f"GrB_Vector_{stem}_{op.kind}")
        check(fn(out._h, mh, ah, C.c_void_p(op.get_op()), self._h, other._h, dh), out)
        return out

    def eadd(self, other, add_op=None, cast=None, out=None, mask=None, accum=None, desc=None):
        return self._ewise("eWiseAdd", other, add_op, cast, out, mask, accum, desc, lambda t: t._default_addop())

    def emult(self, other, mult_op=None, cast=None, out=None, mask=None, accum=None, desc=None):
        return self._ewise("eWiseMult", other, mult_op, cast, out, mask, accum, desc, lambda t: t._default_multop())

    def iseq(self, other, eq_op=None):
        """Same pattern and equal values (reference: vector.py:188-235: size, nvals, eWiseMult(EQ) into BOOL, nvals, LAND-reduce).  Two vectors of one built-in
        real type compared with that type's own EQ go through the library's one-pass form of exactly that (`GrBX_Vector_iseq`)."""
        if self.size != other.size:
            return False
        if eq_op is None and self.type == other.type:
            r = C.c_bool(False)
            info = lib.GrBX_Vector_iseq(C.byref(r), self._h, other._h)
            if info == 0:
                return bool(r.value)
            if info != 1:                       # (1 = GrB_NO_VALUE: not the one-pass case — composed below)
                check(info, self)
        if self.nvals != other.nvals:
            return False
        if eq_op is None:
            eq_op = types.promote(self.type, other.type).EQ
        c = self.emult(other, eq_op, cast=types.BOOL)
        if c.nvals != self.nvals:
            return False
        return c.reduce_bool(types.BOOL.LAND_MONOID)

    def apply(self, op, out=None, mask=None, accum=None, desc=None):
        if out is None:
            out = Vector.sparse(self.type, self.size)
        mh, ah, dh = get_args(mask, accum, desc)
        check(lib.GrB_Vector_apply(out._h, mh, ah, C.c_void_p(op.get_op()), self._h, dh), out)
        return out

    def apply_second(self, op, second, out=None, mask=None, accum=None, desc=None):
        if out is None:
            out = Vector.sparse(self.type, self.size)
        mh, ah, dh = get_args(mask, accum, desc)
        fn = getattr(lib, "GxB_Vector_apply_BinaryOp2nd_" + self.type.__name__)
        check(fn(out._h, mh, ah, C.c_void_p(op.get_op()), self._h, self.type._c(second), dh), out)
        return out

    def apply_first(self, first, op, out=None, mask=None, accum=None, desc=None):
        if out is None:
            out = Vector.sparse(self.type, self.size)
        mh, ah, dh = get_args(mask, accum, desc)
        fn = getattr(lib, "GxB_Vector_apply_BinaryOp1st_" + self.type.__name__)
        check(fn(out._h, mh, ah, C.c_void_p(op.get_op()), self.type._c(first), self._h, dh), out)
        return out

    def select(self, op, thunk=None, out=None, mask=None, accum=None, desc=None):
        """`GxB_Vector_select` with a built-in select operator name ("NONZERO", ">", ...) (reference: pygraphblas/vector.py:1354-1404)."""
        opname = {"nonzero": "NONZERO", "!=0": "NONZERO", "==0": "EQ_ZERO", ">0": "GT_ZERO", ">=0": "GE_ZERO", "<0": "LT_ZERO", "<=0": "LE_ZERO",
                  "!=": "NE_THUNK", "==": "EQ_THUNK", ">": "GT_THUNK", ">=": "GE_THUNK", "<": "LT_THUNK", "<=": "LE_THUNK"}.get(op, op)
        if out is None:
            out = Vector.sparse(self.type, self.size)
        th = None
        if thunk is not None:
            th = C.c_void_p()
            check(lib.GxB_Scalar_new(C.byref(th), C.c_void_p(self.type._h)))
            check(getattr(lib, "GxB_Scalar_setElement_" + self.type.__name__)(th, self.type._c(thunk)))
        mh, ah, dh = get_args(mask, accum, desc)
        try:
            check(lib.GxB_Vector_select(out._h, mh, ah, C.c_void_p(_capi.handle("GxB_" + opname)), self._h, th, dh), out)
        finally:
            if th is not None:
                lib.GxB_Scalar_free(C.byref(th))
        return out

    def nonzero(self):
        return self.select("NONZERO")

    def pattern(self, typ=types.BOOL):
        """The structure of the vector as a vector of ones (reference: pygraphblas/vector.py:1406-1425)."""
        out = Vector.sparse(typ, self.size)
        check(lib.GrB_Vector_apply(out._h, None, None, C.c_void_p(typ.ONE.get_op()), self._h, None), out)
        return out

    def cast(self, cast, out=None):
        if out is None:
            out = Vector.sparse(cast, self.size)
        check(lib.GrB_Vector_apply(out._h, None, None, C.c_void_p(cast.IDENTITY.get_op()), self._h, None), out)
        return out

    def __invert__(self):
        return self.apply(self.type.MINV)

    # ---- operators (reference: pygraphblas/vector.py:982-1076): vector operands -> eadd / emult, scalars -> bound apply
    def _binop(self, other, opname, union, out=None, reverse=False):
        op = getattr(self.type, opname)
        if isinstance(other, Vector):
            return (self.eadd if union else self.emult)(other, op, out=out)
        return self.apply_first(other, op, out=out) if reverse else self.apply_second(op, other, out=out)

    def __add__(self, other):
        return self._binop(other, "PLUS", True)

    def __radd__(self, other):
        return self._binop(other, "PLUS", True, reverse=True)

    def __iadd__(self, other):
        return self._binop(other, "PLUS", True, out=self)

    def __sub__(self, other):
        return self._binop(other, "MINUS", True)

    def __rsub__(self, other):
        return self._binop(other, "MINUS", True, reverse=True)

    def __isub__(self, other):
        return self._binop(other, "MINUS", True, out=self)

    def __mul__(self, other):
        return self._binop(other, "TIMES", False)

    def __rmul__(self, other):
        return self._binop(other, "TIMES", False, reverse=True)

    def __imul__(self, other):
        return self._binop(other, "TIMES", False, out=self)

    def __truediv__(self, other):
        return self._binop(other, "DIV", False)

    def __rtruediv__(self, other):
        return self._binop(other, "DIV", False, reverse=True)

    def __itruediv__(self, other):
        return self._binop(other, "DIV", False, out=self)

    def __abs__(self):
        return self.apply(self.type.ABS)

    def __neg__(self):
        return self.apply(self.type.AINV)

    def __repr__(self):
        return f"<Vector ({self.size} : {self.nvals}:{self.type.__name__})>"
