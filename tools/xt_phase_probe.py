#!/usr/bin/env python3
"""Measurement harness: per-wave cycles of kernel X's tile pipeline by phase (needs the XT_PROFILE build:
  make -C pygraphblas_amd/csrc BUILD=build_prof LIB=../libgrb_prof.so XTFLAGS=-DXT_PROFILE ; GRB_MI355X_LIB=.../libgrb_prof.so python tools/xt_phase_probe.py)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("GRB_MI355X_LIB", os.path.join(ROOT, "pygraphblas_amd", "libgrb_prof.so"))
import numpy as np, torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat
S = 22; dev = torch.device("cuda", 0); n = 1 << S
rowptr, col = rmat.csr_torch(S, dev, seed=42); nnz = int(col.numel())
lib = gb.lib
names = ["until products exist", "issue next loads", "scan + end flags", "staging + stores"]
for tname, ctype, sr in (("FP64", "double", "PLUS_TIMES"), ("FP32", "float", "PLUS_SECOND")):
    typ = getattr(gb, tname)
    vals = rmat.values_torch(nnz, dev, seed=43, dtype=torch.float64 if tname == "FP64" else torch.float32)
    xs = rmat.values_torch(n, dev, seed=44, dtype=torch.float64 if tname == "FP64" else torch.float32)
    A = gb.Matrix.from_csr(typ, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    x = gb.Vector.from_dense_array((xs.data_ptr(), n), typ, device=True); w = gb.Vector.sparse(typ, n)
    for _ in range(5): A.mxv(x, semiring=getattr(typ, sr), out=w)
    out = (C.c_ulonglong * (4096 * 8))()
    getattr(lib, "GrBX_xt_prof_read_" + ctype)(out)
    a = np.frombuffer(out, dtype=np.uint64).astype(np.float64).reshape(4096, 8)
    live = a[:, 4] > 0; a = a[live]
    tiles = a[:, 4]; tot = a[:, 5]
    print(f"{tname} {sr}: waves {live.sum()}  tiles per wave mean {tiles.mean():.1f}  kernel cycles per wave mean {tot.mean():.0f} max {tot.max():.0f}  [{gb.last_kernel_plan()}]")
    per_tile = a[:, :4].sum(0) / tiles.sum()
    for k in range(4):
        print(f"   {names[k]:24s} {per_tile[k]:8.0f} cycles per tile  ({100 * a[:, k].sum() / tot.sum():.1f} % of the waves' time)")
    print(f"   accounted {100 * a[:, :4].sum() / tot.sum():.1f} %; cycles per tile per wave {tot.sum() / tiles.sum():.0f}")
