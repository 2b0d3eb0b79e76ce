"""`suitesparse_graphblas.io.binary` for the MI355X shim: the SuiteSparse / LAGraph ".grb" reader and writer that the
reference's `Matrix.binread` / `binwrite` delegate to (pygraphblas/matrix.py:489-497, 935-942) and its gap/ drivers load
their graphs with (gap/prmark.py:42-48, gap/bcmark.py:77-83).

Layout (little endian), pinned to the reference's fixture docs/test_binfile.grb (see pygraphblas_amd/matrix.py, binread):
  512-byte ASCII header | int32 format | int32 sparsity | f64 hyper_switch | u64 nrows | u64 ncols | i64 nonempty | u64 nvec |
  u64 nvals | int32 typecode | u64 typesize | then Ap/Ah/Ai/Ax (hypersparse, sparse) or Ab/Ax (bitmap) or Ax (full).
`binread` returns a `GrB_Matrix*` cdata, as the original does; the entries go to the backend in one GrB_Matrix_build.
"""
import struct
from pathlib import Path

import numpy as np

from .. import ffi, lib

_TYPES = ["BOOL", "INT8", "INT16", "INT32", "INT64", "UINT8", "UINT16", "UINT32", "UINT64", "FP32", "FP64"]
_NP = {"BOOL": np.bool_, "INT8": np.int8, "INT16": np.int16, "INT32": np.int32, "INT64": np.int64, "UINT8": np.uint8,
       "UINT16": np.uint16, "UINT32": np.uint32, "UINT64": np.uint64, "FP32": np.float32, "FP64": np.float64}
_C = {"BOOL": "_Bool", "INT8": "int8_t", "INT16": "int16_t", "INT32": "int32_t", "INT64": "int64_t", "UINT8": "uint8_t",
      "UINT16": "uint16_t", "UINT32": "uint32_t", "UINT64": "uint64_t", "FP32": "float", "FP64": "double"}


def _check(info):
    if info != lib.GrB_SUCCESS:
        raise RuntimeError(f"GrB_Info {info}")


def binread(filename, opener=Path.open):
    with opener(Path(filename), "rb") as f:
        raw = f.read()
    if len(raw) < 512 + 68 or not raw.startswith(b"SuiteSparse:GraphBLAS matrix"):
        raise ValueError("not a SuiteSparse:GraphBLAS binary matrix file")
    fmt, kind, _hs, nrows, ncols, _nonempty, nvec, nvals, tcode, tsize = struct.unpack_from("<iidQQqQQiQ", raw, 512)
    if not 0 <= tcode < len(_TYPES):
        raise TypeError(f"type code {tcode} is not supported by the MI355X backend")
    name = _TYPES[tcode]; dt = np.dtype(_NP[name])
    if dt.itemsize != tsize:
        raise ValueError("type size in the file does not match its type code")
    pos = [512 + 68]

    def take(dtype, count):
        a = np.frombuffer(raw, dtype=dtype, count=count, offset=pos[0]); pos[0] += a.nbytes
        return a
    nmajor, nminor = (nrows, ncols) if fmt == 0 else (ncols, nrows)
    if kind in (1, 2):
        Ap = take("<u8", nvec + 1)
        Ah = take("<u8", nvec) if kind == 1 else np.arange(nvec, dtype=np.uint64)
        Ai = take("<u8", nvals); Ax = take(dt, nvals)
        major = np.repeat(Ah, np.diff(Ap.astype(np.int64))); minor = Ai
    elif kind in (4, 8):
        Ab = take("i1", nmajor * nminor) if kind == 4 else np.ones(nmajor * nminor, np.int8)
        Ax = take(dt, nmajor * nminor)
        flat = np.flatnonzero(Ab).astype(np.uint64)
        major, minor = np.divmod(flat, np.uint64(max(nminor, 1))); Ax = Ax[flat.astype(np.int64)]
    else:
        raise ValueError(f"unknown sparsity code {kind} in .grb file")
    I, J = (major, minor) if fmt == 0 else (minor, major)
    I = np.ascontiguousarray(I, np.uint64); J = np.ascontiguousarray(J, np.uint64); X = np.ascontiguousarray(Ax, dt)
    A = ffi.new("GrB_Matrix*")
    _check(lib.GrB_Matrix_new(A, getattr(lib, "GrB_" + name), nrows, ncols))
    if len(I):
        _check(getattr(lib, "GrB_Matrix_build_" + name)(A[0], ffi.cast("GrB_Index*", ffi.from_buffer(I)), ffi.cast("GrB_Index*", ffi.from_buffer(J)),
                                                        ffi.cast(_C[name] + "*", ffi.from_buffer(X)), len(I), getattr(lib, "GrB_SECOND_" + name)))
    return A


def binwrite(A, filename, comments=None, opener=Path.open):
    """`A`: GrB_Matrix* cdata.  Written sparse, by row."""
    n = ffi.new("GrB_Index*"); t = ffi.new("GrB_Type*")
    _check(lib.GrB_Matrix_nrows(n, A[0])); nrows = n[0]
    _check(lib.GrB_Matrix_ncols(n, A[0])); ncols = n[0]
    _check(lib.GrB_Matrix_nvals(n, A[0])); nvals = n[0]
    _check(lib.GxB_Matrix_type(t, A[0]))
    name = next(nm for nm in _TYPES if getattr(lib, "GrB_" + nm) == t[0]); dt = np.dtype(_NP[name])
    I = np.zeros(max(nvals, 1), np.uint64); J = np.zeros(max(nvals, 1), np.uint64); X = np.zeros(max(nvals, 1), dt)
    _check(getattr(lib, "GrB_Matrix_extractTuples_" + name)(ffi.cast("GrB_Index*", ffi.from_buffer(I)), ffi.cast("GrB_Index*", ffi.from_buffer(J)),
                                                            ffi.cast(_C[name] + "*", ffi.from_buffer(X)), n, A[0]))
    I, J, X = I[:nvals], J[:nvals], X[:nvals]
    order = np.lexsort((J, I)); I, J, X = I[order], J[order], X[order]
    rp = np.zeros(nrows + 1, np.uint64); np.cumsum(np.bincount(I.astype(np.int64), minlength=nrows), out=rp[1:])
    head = (f"SuiteSparse:GraphBLAS matrix\nv4.0.1 (LAGraph DRAFT)\nnrows:  {nrows}\nncols:  {ncols}\nnvec:   {nrows}\nnvals:  {nvals}\n"
            f"format: SPARSER\nsize:   {dt.itemsize}\ntype:   GrB_{name}\n{comments or ''}\n").encode()
    blob = head[:511].ljust(511, b" ") + b"\0"
    blob += struct.pack("<iidQQqQQiQ", 0, 2, 0.0625, nrows, ncols, -1, nrows, nvals, _TYPES.index(name), dt.itemsize)
    blob += rp.astype("<u8").tobytes() + J.astype("<u8").tobytes() + X.tobytes()
    with opener(Path(filename), "wb") as f:
        f.write(blob)
