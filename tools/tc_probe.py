#!/usr/bin/env python3
"""Measurement harness (not part of the product): one masked SpGEMM (triangle count L.mxm(L, PLUS_PAIR, mask=L)) on R-MAT for
profiling runs.  --serial runs the five row bins one after the other on the library stream (counter collection needs that)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--scale", type=int, default=22); ap.add_argument("--reps", type=int, default=1); ap.add_argument("--serial", action="store_true")
args = ap.parse_args()
if args.serial:
    os.environ["GRB_MI355X_SPGEMM_SERIAL"] = "1"
import torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat
dev = torch.device("cuda", 0); n = 1 << args.scale
rowptr, col = rmat.csr_torch(args.scale, dev, seed=42, symmetric=True, drop_self_loops=True, lower=True)
nnz = int(col.numel()); vals = torch.ones(nnz, dtype=torch.int64, device=dev)
L = gb.Matrix.from_csr(gb.INT64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
dL = (rowptr[1:] - rowptr[:-1]).to(torch.int64); flops = 2 * int(dL[col.to(torch.int64) & 0xFFFFFFFF].sum())
best = 1e9
for _ in range(args.reps):
    torch.cuda.synchronize(); t = time.perf_counter(); tri = L.mxm(L, semiring=gb.INT64.PLUS_PAIR, mask=L).reduce_int(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
alg = 2 * (nnz * 4 + (n + 1) * 4) + (flops // 2) * 4 + nnz * 12
print(json.dumps({"scale": args.scale, "nnz_L": nnz, "triangles": int(tri), "flops": flops, "seconds": round(best, 5), "algorithmic_bytes": alg, "serial": args.serial, "plan": gb.last_kernel_plan()}))
