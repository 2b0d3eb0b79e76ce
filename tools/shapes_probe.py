#!/usr/bin/env python3
"""Experiment harness (not part of the product): kernel X / W on matrices that are not R-MAT — uniform random, banded, a few
dense columns, a few dense rows — checked against scipy.sparse, with timings."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp, torch
import pygraphblas_amd as gb
dev = torch.device("cuda", 0); lib = gb.lib
rng = np.random.default_rng(5)
n = 1 << 20


def rand_csr(per_row, seed):
    """about per_row uniformly placed entries per row (numpy coordinates; scipy.sparse.random samples without replacement from
    n*n positions and exhausts the memory of the machine at this size)"""
    r = np.random.default_rng(seed); m = per_row * n
    i = r.integers(0, n, m); j = r.integers(0, n, m)
    A = sp.csr_matrix((r.random(m), (i, j)), shape=(n, n)); A.sum_duplicates(); return A


def uniform():
    return rand_csr(16, 1)


def banded():
    offs = list(range(-8, 8)); return sp.diags([rng.random(n - abs(o)) for o in offs], offs, shape=(n, n), format="csr")


def dense_cols():
    A = rand_csr(8, 2)
    cols = rng.choice(n, 8, replace=False)
    B = sp.csr_matrix((rng.random(8 * n), (np.tile(np.arange(n), 8), np.repeat(cols, n))), shape=(n, n))
    return (A + B).tocsr()


def dense_rows():
    A = rand_csr(8, 3)
    rows = rng.choice(n, 8, replace=False)
    B = sp.csr_matrix((rng.random(8 * n), (np.repeat(rows, n), np.tile(np.arange(n), 8))), shape=(n, n))
    return (A + B).tocsr()


for name, make in (("uniform", uniform), ("banded", banded), ("dense_cols", dense_cols), ("dense_rows", dense_rows)):
    S = make(); S.sort_indices(); nnz = S.nnz
    rp = torch.from_numpy(S.indptr.astype(np.int32)).to(dev); ci = torch.from_numpy(S.indices.astype(np.int32)).to(dev)
    av = torch.from_numpy(S.data).to(dev)
    xv = rng.random(n); xd = torch.from_numpy(xv).to(dev)
    A = gb.Matrix.from_csr(gb.FP64, n, n, rp.data_ptr(), ci.data_ptr(), (av.data_ptr(), nnz), device=True)
    x = gb.Vector.from_dense_array((xd.data_ptr(), n), gb.FP64, device=True); w = gb.Vector.sparse(gb.FP64, n)
    exp = S @ xv; nonempty = np.diff(S.indptr) > 0
    for method in ("auto", "xcd", "wavepipe", "adaptive"):
        os.environ["GRB_MI355X_SPMV"] = method
        for _ in range(3): A.mxv(x, semiring=gb.FP64.PLUS_TIMES, out=w)
        gv, gp = w.to_dense_arrays()
        ok = np.array_equal(gp.astype(bool), nonempty) and np.allclose(gv[nonempty], exp[nonempty], rtol=1e-9)
        torch.cuda.synchronize(); lib.GrBX_timer_start()
        for _ in range(20): A.mxv(x, semiring=gb.FP64.PLUS_TIMES, out=w)
        ms = C.c_float(0); lib.GrBX_timer_stop(C.byref(ms)); t = ms.value / 20
        print(f"{name:11s} nnz {nnz:9d} {method:9s} {t:7.4f} ms {2*nnz/t/1e6:7.1f} GFLOP/s  {'ok' if ok else 'MISMATCH'}  [{gb.last_kernel_plan().strip()}]", flush=True)
