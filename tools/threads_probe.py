#!/usr/bin/env python3
"""Two host threads drive the library at once, each on its own matrices and vectors (ctypes releases the GIL during the calls): BFS levels and PageRank vectors
must equal the single-threaded ones.  usage: python tools/threads_probe.py [--scale 18] [--rounds 3]"""
import argparse, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat, loops

ap = argparse.ArgumentParser(); ap.add_argument("--scale", type=int, default=18); ap.add_argument("--rounds", type=int, default=3); args = ap.parse_args()
dev = torch.device("cuda", 0); S = args.scale; n = 1 << S

def make(seed):
    rowptr, col = rmat.csr_torch(S, dev, seed=seed, symmetric=True, drop_self_loops=True)
    nnz = int(col.numel())
    ones = torch.ones(nnz, dtype=torch.bool, device=dev)
    A = gb.Matrix.from_csr(gb.BOOL, n, n, rowptr.data_ptr(), col.data_ptr(), (ones.data_ptr(), nnz), device=True)
    f = torch.ones(nnz, dtype=torch.float32, device=dev)
    P = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (f.data_ptr(), nnz), device=True)
    src = int(torch.argmax(rowptr[1:] - rowptr[:-1]))
    return A, P, src, (rowptr, col, ones, f)

def work(A, P, src):
    v, depth = loops.bfs(A, src)
    lv = v.to_dense_arrays()
    d = P.reduce_vector()
    r, its = loops.pagerank(P, d)[:2]
    return lv[0].copy(), lv[1].copy(), depth, r.to_dense_arrays()[0].copy(), its

jobs = [make(42), make(43)]
ref = [work(A, P, s) for A, P, s, _ in jobs]
bad = 0
for rnd in range(args.rounds):
    out = [None, None]; err = [None, None]
    def run(k):
        try: out[k] = work(*jobs[k][:3])
        except Exception as e: err[k] = e
    ts = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    for k in range(2):
        if err[k] is not None: print("round", rnd, "thread", k, "raised", type(err[k]).__name__, err[k]); bad += 1; continue
        a, b = ref[k], out[k]
        ok = np.array_equal(a[0][a[1] != 0], b[0][b[1] != 0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[4] == b[4] and np.allclose(a[3], b[3], rtol=1e-5, atol=1e-9)
        if not ok: print("round", rnd, "thread", k, "differs: depth", a[2], b[2], "iterations", a[4], b[4]); bad += 1
print("two threads:", "ok" if bad == 0 else f"{bad} problems", f"({args.rounds} rounds, R-MAT-{S})")
sys.exit(1 if bad else 0)
