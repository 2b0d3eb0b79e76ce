#!/usr/bin/env python3
"""Experiment: does confining the gathers of one launch to a 1/P column window of x remove the L2 misses?
Times P launches (one per column panel) of the existing kernels on the panel sub-matrices."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat
ap = argparse.ArgumentParser(); ap.add_argument("--scale", type=int, default=22); ap.add_argument("--panels", type=int, default=8)
ap.add_argument("--method", default="wavepipe"); ap.add_argument("--by", default="range", help="range|rank")
args = ap.parse_args()
dev = torch.device("cuda", 0); n = 1 << args.scale; lib = gb.lib
rowptr, col = rmat.csr_torch(args.scale, dev, seed=42)
nnz = col.numel(); cl = col.to(torch.int64) & 0xFFFFFFFF
rows = torch.repeat_interleave(torch.arange(n, device=dev), (rowptr[1:] - rowptr[:-1]).to(torch.int64))
vals = rmat.values_torch(nnz, dev, seed=43); xs = rmat.values_torch(n, dev, seed=44)
if args.by == "rank":
    cnt = torch.bincount(cl, minlength=n); order = torch.argsort(cnt, descending=True, stable=True)
    rank = torch.empty_like(order); rank[order] = torch.arange(n, device=dev)
    panel_of = rank % args.panels
else:
    panel_of = cl // (n // args.panels)
    panel_of = None
x = gb.Vector.from_dense_array((xs.data_ptr(), n), gb.FP64, device=True)
mats = []
for p in range(args.panels):
    sel = (rank[cl] % args.panels == p) if args.by == "rank" else (cl // (n // args.panels) == p)
    r = rows[sel]; c = cl[sel].to(torch.int32); v = vals[sel]
    rp = torch.zeros(n + 1, dtype=torch.int64, device=dev); rp[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
    rp = rp.to(torch.int32)
    mats.append((gb.Matrix.from_csr(gb.FP64, n, n, rp.data_ptr(), c.data_ptr(), (v.data_ptr(), c.numel()), device=True), c.numel(), gb.Vector.sparse(gb.FP64, n)))
os.environ["GRB_MI355X_SPMV"] = args.method
for A, k, w in mats:
    for _ in range(2): A.mxv(x, semiring=gb.FP64.PLUS_TIMES, out=w)
torch.cuda.synchronize(); lib.GrBX_timer_start()
reps = 10
for _ in range(reps):
    for A, k, w in mats: A.mxv(x, semiring=gb.FP64.PLUS_TIMES, out=w)
ms = C.c_float(0); lib.GrBX_timer_stop(C.byref(ms))
print(f"panels={args.panels} by={args.by} method={args.method}: {ms.value/reps:.4f} ms for all panels; entries per panel {[k for _,k,_ in mats]}")
