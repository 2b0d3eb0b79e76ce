import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, numpy as np
import pygraphblas_amd as gb
from pygraphblas_amd import rmat, descriptor as D
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << S; dev = torch.device("cuda", 0)
rowptr, col = rmat.csr_torch(S, dev, seed=42, drop_self_loops=True)
nnz = col.numel(); vals = torch.ones(nnz, dtype=torch.float32, device=dev)
A = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
def T(label, f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); print(f"{label:40s} {time.perf_counter()-t:8.4f} s", flush=True); return r
AT = T("A.transpose()", lambda: A.transpose())
ns = 4; deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
sources = [int(x) for x in torch.argsort(deg, descending=True, stable=True)[:ns].cpu()]
Matrix, Vector, FP32, BOOL = gb.Matrix, gb.Vector, gb.FP32, gb.BOOL
paths = T("Matrix.dense(ns,n,0)", lambda: Matrix.dense(FP32, ns, n, 0))
frontier = Matrix.sparse(FP32, ns, n)
def seed():
    for i, s in enumerate(sources):
        paths[i, sources[i]] = 1; frontier[i, sources[i]] = 1
T("setElement x8", seed)
T("first mxm", lambda: frontier.mxm(A, out=frontier, mask=paths, semiring=FP32.PLUS_FIRST, desc=D.RC))
Sl = []
for depth in range(n):
    nv = T("frontier.nvals", lambda: frontier.nvals)
    if nv == 0: break
    s = Matrix.sparse(BOOL, ns, n)
    T("apply ONE", lambda: frontier.apply(BOOL.ONE, out=s)); Sl.append(s)
    T("paths.assign_matrix accum", lambda: paths.assign_matrix(frontier, accum=FP32.PLUS))
    T("mxm fwd", lambda: frontier.mxm(A, out=frontier, mask=paths, semiring=FP32.PLUS_FIRST, desc=D.RC))
bcu = T("Matrix.dense(ns,n,1)", lambda: Matrix.dense(FP32, ns, n, 1))
W = Matrix.sparse(FP32, ns, n)
for i in range(depth - 1, 0, -1):
    T("emult DIV mask R", lambda: bcu.emult(paths, FP32.DIV, out=W, mask=Sl[i], desc=D.R))
    T("mxm bwd", lambda: W.mxm(AT, out=W, mask=Sl[i - 1], semiring=FP32.PLUS_FIRST, desc=D.R))
    T("emult TIMES accum", lambda: W.emult(paths, FP32.TIMES, out=bcu, accum=FP32.PLUS))
cent = T("Vector.dense", lambda: Vector.dense(FP32, n, -ns))
T("reduce_vector T0", lambda: bcu.reduce_vector(accum=FP32.PLUS, out=cent, desc=D.T0))
