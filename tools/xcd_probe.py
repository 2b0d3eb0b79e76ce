#!/usr/bin/env python3
"""Does the workgroup -> XCD deal survive other work in the process?  PageRank iterations (kernel X) timed before and after a masked
SpGEMM that used the five side streams; the XCC_ID probe read at both points (GRB_MI355X_XCD_REPROBE=1 makes every call probe anew)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat, loops
S = 22; n = 1 << S; dev = torch.device("cuda", 0)
rowptr, col = rmat.csr_torch(S, dev, seed=42); nnz = int(col.numel())
ones = torch.ones(nnz, dtype=torch.float32, device=dev)
A = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (ones.data_ptr(), nnz), device=True)
deg = (rowptr[1:] - rowptr[:-1]).to(torch.float32); pres = (deg > 0).to(torch.uint8)
def d(): return gb.Vector.from_dense_array((deg.data_ptr(), n), gb.FP32, present=pres.data_ptr(), device=True)
def probe():
    b = C.create_string_buffer(128); gb.lib.GrBX_xcd_mapping(b, 128); return b.value.decode()
def pr(tag):
    loops.pagerank(A, d(), fixed_iterations=3)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); loops.pagerank(A, d(), fixed_iterations=20); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print(f"{tag}: {best / 20 * 1e3:.4f} ms per iteration; probe: {probe()}", flush=True)
pr("fresh process")
rp2, c2 = rmat.csr_torch(20, dev, seed=42, symmetric=True, drop_self_loops=True, lower=True)
v2 = torch.ones(int(c2.numel()), dtype=torch.int64, device=dev)
L = gb.Matrix.from_csr(gb.INT64, 1 << 20, 1 << 20, rp2.data_ptr(), c2.data_ptr(), (v2.data_ptr(), int(c2.numel())), device=True)
print("triangles", L.mxm(L, semiring=gb.INT64.PLUS_PAIR, mask=L).reduce_int())
pr("after a masked SpGEMM on five streams")
