"""Offline stand-in for `mmparse` (test harness), the Matrix Market reader the reference's `Matrix.from_mm` imports (pygraphblas/matrix.py:395-409):
`with mmread(path) as (header, rows)` where header has mm_type / mm_storage / nrows / ncols and rows yields (line number, i, j, value) with 0-based
indices; `get_mm_type_converter(mm_type)` names the GraphBLAS type of the file's field.  Coordinate format only (what the notebooks and docs/test_mm.mm use)."""
from contextlib import contextmanager


def get_mm_type_converter(mm_type):
    from pygraphblas import types
    return {"pattern": types.BOOL, "integer": types.INT64, "real": types.FP64, "double": types.FP64}[mm_type]


@contextmanager
def mmread(path):
    f = open(path, "r")
    try:
        banner = f.readline().split()
        if len(banner) < 5 or banner[0] != "%%MatrixMarket" or banner[2] != "coordinate":
            raise ValueError("mmparse stand-in: only `%%MatrixMarket matrix coordinate ...` files")
        mm_type, mm_storage = banner[3].lower(), banner[4].lower()
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        nrows, ncols, nnz = (int(x) for x in line.split())
        conv = {"pattern": lambda t: True, "integer": lambda t: int(t[2]), "real": lambda t: float(t[2]), "double": lambda t: float(t[2])}[mm_type]

        def rows():
            for k, ln in enumerate(f):
                t = ln.split()
                if t:
                    yield k, int(t[0]) - 1, int(t[1]) - 1, conv(t)
        yield dict(mm_type=mm_type, mm_storage=mm_storage, nrows=nrows, ncols=ncols, nnz=nnz), rows()
    finally:
        f.close()
