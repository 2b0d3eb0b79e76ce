import os, sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat
dev = torch.device("cuda", 0); S = 22; n = 1 << S
rowptr, col = rmat.csr_torch(S, dev, seed=42); nnz = int(col.numel())
vals = rmat.values_torch(nnz, dev, seed=43); xs = rmat.values_torch(n, dev, seed=44)
x = gb.Vector.from_dense_array((xs.data_ptr(), n), gb.FP64, device=True)
lib = gb.lib
os.environ["GRB_MI355X_SPMV"] = "wavepipe"
for samp in (1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 26):
    os.environ["GRB_MI355X_WP_SAMPLE"] = str(samp)
    A = gb.Matrix.from_csr(gb.FP64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    w = gb.Vector.sparse(gb.FP64, n)
    torch.cuda.synchronize(); t0 = time.perf_counter(); A.mxv(x, semiring=gb.FP64.PLUS_TIMES, out=w); torch.cuda.synchronize(); first = (time.perf_counter() - t0) * 1e3
    for _ in range(3): A.mxv(x, semiring=gb.FP64.PLUS_TIMES, out=w)
    torch.cuda.synchronize(); lib.GrBX_timer_start()
    for _ in range(30): A.mxv(x, semiring=gb.FP64.PLUS_TIMES, out=w)
    ms = C.c_float(0); lib.GrBX_timer_stop(C.byref(ms)); t = ms.value / 30
    alg = nnz * 12 + (n + 1) * 4 + 2 * n * 8
    print(f"sample {samp:9d}: first call {first:6.2f} ms, steady {t:.4f} ms, frac {alg / t / 1e6 / 8000:.4f}", flush=True)
