#!/usr/bin/env python3
"""Experiment harness (not part of the product): bit-exactness of the panel kernel on integer / boolean semirings at scale
(scipy.sparse as the checker)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp, torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 19
dev = torch.device("cuda", 0); n = 1 << scale
rowptr, col = rmat.csr_torch(scale, dev, seed=42); nnz = col.numel()
rp = rowptr.cpu().numpy().view(np.uint32).astype(np.int64); ci = col.cpu().numpy().view(np.uint32).astype(np.int64)
rng = np.random.default_rng(1)
os.environ["GRB_MI355X_SPMV"] = "xcd"
ok = True
for tname, npdt, lo, hi in (("INT64", np.int64, -3, 4), ("INT32", np.int32, -3, 4), ("UINT8", np.uint8, 0, 3)):
    T = getattr(gb, tname)
    av = rng.integers(lo, hi, nnz).astype(npdt); xv = rng.integers(lo, hi, n).astype(npdt)
    avd = torch.from_numpy(av).to(dev); xvd = torch.from_numpy(xv).to(dev)
    A = gb.Matrix.from_csr(T, n, n, rowptr.data_ptr(), col.data_ptr(), (avd.data_ptr(), nnz), device=True)
    x = gb.Vector.from_dense_array((xvd.data_ptr(), n), T, device=True); w = gb.Vector.sparse(T, n)
    S = sp.csr_matrix((av.astype(np.int64), ci, rp), shape=(n, n))
    for srn in ("PLUS_TIMES", "MIN_PLUS", "PLUS_PAIR"):
        A.mxv(x, semiring=getattr(T, srn), out=w)
        gv, gp = w.to_dense_arrays()
        rows_nonempty = np.diff(rp) > 0
        if srn == "PLUS_TIMES":
            exp = (S @ xv.astype(np.int64)).astype(npdt)          # wraps like the device type
        elif srn == "PLUS_PAIR":
            exp = np.diff(rp).astype(npdt)
        else:
            big = np.iinfo(npdt).max
            vals = (av.astype(np.int64) + xv.astype(np.int64)[ci]).astype(npdt)
            exp = np.full(n, big, npdt); np.minimum.at(exp, np.repeat(np.arange(n), np.diff(rp)), vals)
        good = np.array_equal(gp.astype(bool), rows_nonempty) and np.array_equal(gv[rows_nonempty], exp[rows_nonempty])
        ok &= good
        print(f"{tname}.{srn}: {'bit-exact' if good else 'MISMATCH'}  [{gb.last_kernel_plan()}]", flush=True)
print("ALL OK" if ok else "FAILED")
