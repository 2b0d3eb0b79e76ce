#!/usr/bin/env python3
"""Experiment harness (not part of the product): per-wave cycle counts of the SpMV wave pipeline.
Needs grb_spmv_inst.hip compiled with -DWP_PROFILE for double (exports GrBX_wp_prof_read)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
dev = torch.device("cuda", 0); n = 1 << scale
rowptr, col = rmat.csr_torch(scale, dev, seed=42); nnz = col.numel()
vals = rmat.values_torch(nnz, dev, seed=43); xs = rmat.values_torch(n, dev, seed=44)
A = gb.Matrix.from_csr(gb.FP64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
x = gb.Vector.from_dense_array((xs.data_ptr(), n), gb.FP64, device=True); w = gb.Vector.sparse(gb.FP64, n)
lib = C.CDLL(os.path.join(ROOT, "pygraphblas_amd", "libgrb_mi355x.so"))
out = (C.c_ulonglong * 8192)()
for method in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["xcd", "wavepipe"]):
    os.environ["GRB_MI355X_SPMV"] = method
    for srn in ("PLUS_TIMES", "PLUS_PAIR"):
        for _ in range(5): A.mxv(x, semiring=getattr(gb.FP64, srn), out=w)
        lib.GrBX_wp_prof_read(out)
        a = np.frombuffer(out, dtype=np.uint64).astype(np.float64)
        t, k = a[:4096], a[4096:]
        live = k > 0
        t = t[live]; wg = np.arange(4096)[live] // 16
        print(f"{method} {srn}: waves {live.sum()} cycles mean {t.mean():.0f} p50 {np.median(t):.0f} p90 {np.percentile(t,90):.0f} max {t.max():.0f}  max/mean {t.max()/t.mean():.2f}  [{gb.last_kernel_plan()}]")
        wgmax = np.array([t[wg == g].max() for g in np.unique(wg)])
        print("   per-workgroup max: mean %.0f min %.0f max %.0f;  by XCD (wg %% 8) mean of max: %s" % (wgmax.mean(), wgmax.min(), wgmax.max(),
              " ".join("%.0f" % wgmax[np.unique(wg) % 8 == j].mean() for j in range(8))), flush=True)
