"""pygraphblas_amd — MI355X-native GraphBLAS hot path (GrB_mxm / GrB_mxv / GrB_vxm in hand-written HIP)
behind the pygraphblas `Matrix` / `Vector` / semiring surface.

    from pygraphblas_amd import *
    A = Matrix.from_lists([0, 1, 2], [1, 2, 0], [1, 2, 3])
    v = Vector.from_lists([0, 1, 2], [2, 3, 4])
    A @ v                      # GrB_mxv on the GPU
    with INT64.MIN_PLUS: A @ A # GrB_mxm on the GPU

The package is a thin host mirror (ctypes) over `libgrb_mi355x.so`; see DESIGN.md.
"""
from . import _capi
from ._capi import lib

_capi.init()

from .base import (GraphBLASException, NoValue, UninitializedObject, InvalidObject, NullPointer, InvalidValue,  # noqa: E402
                   InvalidIndex, DomainMismatch, DimensionMismatch, OutputNotEmpty, OutOfMemory, InsufficientSpace,
                   IndexOutOfBound, Panic)
from . import types, descriptor  # noqa: E402
from .types import (BOOL, INT8, UINT8, INT16, UINT16, INT32, UINT32, INT64, UINT64, FP32, FP64, Accum, BinaryOp, Monoid,  # noqa: E402
                    Semiring, UnaryOp, promote)
from .matrix import Matrix  # noqa: E402
from .vector import Vector  # noqa: E402

GxB_INDEX_MAX = _capi.constants["GxB_INDEX_MAX"]
device_info = _capi.device_info
last_kernel_plan = _capi.last_kernel_plan

__all__ = ["lib", "Matrix", "Vector", "types", "descriptor", "Accum", "BinaryOp", "Monoid", "Semiring", "UnaryOp", "promote",
           "BOOL", "INT8", "UINT8", "INT16", "UINT16", "INT32", "UINT32", "INT64", "UINT64", "FP32", "FP64",
           "GraphBLASException", "NoValue", "UninitializedObject", "InvalidObject", "NullPointer", "InvalidValue",
           "InvalidIndex", "DomainMismatch", "DimensionMismatch", "OutputNotEmpty", "OutOfMemory", "InsufficientSpace",
           "IndexOutOfBound", "Panic", "GxB_INDEX_MAX", "device_info", "last_kernel_plan"]
