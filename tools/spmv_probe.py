#!/usr/bin/env python3
"""Experiment harness (not part of the product): time GrB_mxv variants on one R-MAT graph resident in HBM."""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--variants", default="FP64.PLUS_TIMES,FP64.PLUS_FIRST,FP64.PLUS_SECOND,FP64.PLUS_PAIR,FP32.PLUS_TIMES,FP32.PLUS_SECOND")
ap.add_argument("--methods", default="adaptive,rowgroup")
ap.add_argument("--loadmodes", default="", help="comma list of g,s pairs like 10,01,21")
ap.add_argument("--relabel", default="none", help="none|degree : relabel columns by popularity (experiment)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
n = 1 << args.scale
rowptr, col = rmat.csr_torch(args.scale, dev, seed=42)
nnz = col.numel()
if args.relabel.startswith("mod"):
    col = ((col.to(torch.int64) & 0xFFFFFFFF) % int(args.relabel[3:])).to(torch.int32)
if args.relabel == "degree":
    cl = col.to(torch.int64) & 0xFFFFFFFF
    cnt = torch.bincount(cl, minlength=n)
    order = torch.argsort(cnt, descending=True, stable=True)
    newid = torch.empty_like(order); newid[order] = torch.arange(n, device=dev)
    col = newid[cl].to(torch.int32)
    del cl, cnt, order, newid
lib = gb.lib
print(f"scale {args.scale} n {n} nnz {nnz}")
mats = {}
for var in args.variants.split(","):
    tname, sr = var.split(".")
    T = getattr(gb, tname)
    if tname not in mats:
        if tname.startswith("INT"):            # (round 6: integer weights 1 ... 255 — the shortest-path problem's — and operand values 0 ... 999)
            it = torch.int64 if tname == "INT64" else torch.int32
            vals = ((rmat.values_torch(nnz, dev, seed=43) * 255.0).to(torch.int64) + 1).to(it)
            xs = (rmat.values_torch(n, dev, seed=44) * 1000.0).to(it)
        else:
            vals = rmat.values_torch(nnz, dev, seed=43, dtype=torch.float64 if tname == "FP64" else torch.float32)
            xs = rmat.values_torch(n, dev, seed=44, dtype=torch.float64 if tname == "FP64" else torch.float32)
        A = gb.Matrix.from_csr(T, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
        x = gb.Vector.from_dense_array((xs.data_ptr(), n), T, device=True)
        mats[tname] = (A, x, gb.Vector.sparse(T, n))
    A, x, w = mats[tname]
    ts = 8 if tname in ("FP64", "INT64") else 4
    runs = [(m, None) for m in args.methods.split(",")] + [("adaptive", lm) for lm in args.loadmodes.split(",") if lm]
    for method, lm in runs:
        os.environ["GRB_MI355X_SPMV"] = method
        os.environ["GRB_MI355X_GATHER"] = lm[0] if lm else "0"
        os.environ["GRB_MI355X_STREAM"] = lm[1] if lm else "0"
        for _ in range(3):
            A.mxv(x, semiring=getattr(T, sr), out=w)
        torch.cuda.synchronize()
        lib.GrBX_timer_start()
        for _ in range(args.reps):
            A.mxv(x, semiring=getattr(T, sr), out=w)
        ms = C.c_float(0); lib.GrBX_timer_stop(C.byref(ms))
        t = ms.value / args.reps
        alg = nnz * (ts + 4) + (n + 1) * 4 + 2 * n * ts
        print(f"{var:18s} {method:9s} {t:8.4f} ms  {2*nnz/t/1e6:8.1f} GFLOP/s  alg {alg/t/1e6:8.1f} GB/s  [{gb.last_kernel_plan()}]", flush=True)
