"""Debug aid: GRB_MI355X_XC_VERIFY=1 python tools/xc_check.py [scale] — kernel X on an R-MAT graph against scipy."""
import sys, os, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
rp, ci = rmat.csr_numpy(scale); n = 1 << scale
rng = np.random.default_rng(1)
val = rng.random(len(ci)) + 0.5
x = rng.random(n) + 0.5
A = gb.Matrix.from_csr(gb.FP64, n, n, rp, ci, val)
u = gb.Vector.from_dense_array(x, gb.FP64)
w = A.mxv(u, semiring=gb.FP64.PLUS_TIMES)
print(gb.last_kernel_plan())
got, pres = w.to_dense_arrays()
S = sp.csr_matrix((val, ci, rp), shape=(n, n)); y = S @ x
nz = np.diff(rp) > 0
bad = np.flatnonzero(~np.isclose(got[nz], y[nz], rtol=1e-9))
print("rows", int(nz.sum()), "bad", len(bad), bad[:10])
