#!/usr/bin/env python3
"""Measurement harness: the O(nnz) matrix companions of the hot path (select / eWise / apply / reduce_vector / transpose / mask
write-back) on R-MAT, where thread-per-row kernels meet hub rows of 1e5 entries."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat
S = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << S; dev = torch.device("cuda", 0)
rowptr, col = rmat.csr_torch(S, dev, seed=42, symmetric=True, drop_self_loops=True)
nnz = col.numel(); vals = rmat.values_torch(nnz, dev, seed=43, dtype=torch.float32)
A = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
def T(label, f, reps=2):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print(f"{label:34s} {best*1e3:10.3f} ms   ({nnz / best / 1e9:7.2f} G entries/s)", flush=True); return r
L = T("A.tril()", lambda: A.tril())
T("A.triu()", lambda: A.triu())
T("A.offdiag()", lambda: A.offdiag())
T("A.select('>0')", lambda: A.select(">0"))
T("A.pattern()", lambda: A.pattern())
T("A.apply(ABS)", lambda: A.apply(gb.FP32.ABS))
T("A.eadd(A)", lambda: A.eadd(A))
T("A.emult(L)", lambda: A.emult(L))
T("A.reduce_vector()", lambda: A.reduce_vector())
T("A.transpose()", lambda: A.transpose())
T("A.reduce_float()", lambda: A.reduce_float())
T("A.dup()", lambda: A.dup())
C = A.dup()
T("C<L> = A (mask write-back)", lambda: A.apply(gb.FP32.IDENTITY, out=C, mask=L))
