#!/usr/bin/env python3
"""Measurement harness (not part of the product): time GrB_mxv FP64 PLUS_TIMES on R-MAT with the tile pipeline of
kernel X in several prefetch-depth / waves-per-workgroup variants (library built with `make XTFLAGS=-DXT_VARIANTS`),
plus the plan-less kernels (A, W) and a label-permuted graph.  Prints one line per measurement."""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--variants", default="d1w16,d1w16e1,d1w16e2,d2w16,d1w8,d2w8,d3w8,d4w8,d2w8e2,d3w8e2")
ap.add_argument("--others", default="wavepipe,adaptive")
ap.add_argument("--permuted", action="store_true")
ap.add_argument("--static", default="", help="comma list of GRB_MI355X_WP_STATIC values to sweep with the default variant (needs a plan rebuild each)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
lib = gb.lib
n = 1 << args.scale


def build(permute_seed=None):
    rowptr, col = rmat.csr_torch(args.scale, dev, seed=42, permute_seed=permute_seed)
    nnz = col.numel()
    vals = rmat.values_torch(nnz, dev, seed=43)
    xs = rmat.values_torch(n, dev, seed=44)
    A = gb.Matrix.from_csr(gb.FP64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    x = gb.Vector.from_dense_array((xs.data_ptr(), n), gb.FP64, device=True)
    return A, x, nnz


def timed(A, x, w, reps):
    for _ in range(3):
        A.mxv(x, semiring=gb.FP64.PLUS_TIMES, out=w)
    torch.cuda.synchronize()
    lib.GrBX_timer_start()
    for _ in range(reps):
        A.mxv(x, semiring=gb.FP64.PLUS_TIMES, out=w)
    ms = C.c_float(0); lib.GrBX_timer_stop(C.byref(ms))
    return ms.value / reps


def report(tag, t, nnz, extra=""):
    alg = nnz * 12 + (n + 1) * 4 + 2 * n * 8
    print(f"{tag:22s} {t:8.4f} ms  {2 * nnz / t / 1e6:8.1f} GFLOP/s  alg {alg / t / 1e6:8.1f} GB/s  frac {alg / t / 1e6 / 8000:6.4f}  {extra}", flush=True)


A, x, nnz = build()
w = gb.Vector.sparse(gb.FP64, n)
print(f"scale {args.scale} n {n} nnz {nnz}", flush=True)
os.environ.pop("GRB_MI355X_XT", None)
t = timed(A, x, w, args.reps)
pb = C.c_float(0); lib.GrBX_last_plan_build_ms(C.byref(pb))
ref, refp = w.to_dense_arrays()
report("default", t, nnz, f"plan_build_ms {pb.value:.2f} [{gb.last_kernel_plan()}]")
for v in [v for v in args.variants.split(",") if v]:
    os.environ["GRB_MI355X_XT"] = v
    t = timed(A, x, w, args.reps)
    extra = ""
    if "e" not in v:
        y, p = w.to_dense_arrays()
        extra = f"same_bits {bool(np.array_equal(y[refp != 0], ref[refp != 0]) and np.array_equal(p, refp))}"
    report(v, t, nnz, extra)
os.environ.pop("GRB_MI355X_XT", None)
for m in [m for m in args.others.split(",") if m]:
    os.environ["GRB_MI355X_SPMV"] = m
    t = timed(A, x, w, args.reps)
    y, p = w.to_dense_arrays()
    ok = bool(np.array_equal(p, refp) and np.allclose(y[refp != 0], ref[refp != 0], rtol=1e-9, atol=0))
    report(m, t, nnz, f"agrees {ok} [{gb.last_kernel_plan()}]")
os.environ.pop("GRB_MI355X_SPMV", None)
for s in [s for s in args.static.split(",") if s]:
    os.environ["GRB_MI355X_WP_STATIC"] = s
    del A, x
    A, x, nnz = build()
    t = timed(A, x, w, args.reps)
    report(f"static_pct={s}", t, nnz)
os.environ.pop("GRB_MI355X_WP_STATIC", None)
if args.permuted:
    del A, x
    A, x, nnz = build(permute_seed=7)
    t = timed(A, x, w, args.reps)
    lib.GrBX_last_plan_build_ms(C.byref(pb))
    report("permuted labels", t, nnz, f"plan_build_ms {pb.value:.2f} [{gb.last_kernel_plan()}]")
    os.environ["GRB_MI355X_SPMV"] = "adaptive"
    t2 = timed(A, x, w, args.reps)
    report("permuted, kernel A", t2, nnz)
    os.environ.pop("GRB_MI355X_SPMV", None)
