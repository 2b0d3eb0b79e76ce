#!/usr/bin/env python3
"""Where a BFS level's time goes: the reference's loop on R-MAT-22 with (a) wall time per mirror call, asynchronous as the loop
runs it, (b) the same with a device synchronize after every call (kernel + launch time of that call alone), (c) the whole loop.
usage: python tools/bfs_probe.py [--scale 22]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat, descriptor as D

ap = argparse.ArgumentParser(); ap.add_argument("--scale", type=int, default=22); ap.add_argument("--only-async", action="store_true"); args = ap.parse_args()
S = args.scale; n = 1 << S; dev = torch.device("cuda", 0)
rowptr, col = rmat.csr_torch(S, dev, seed=42, symmetric=True, drop_self_loops=True)
nnz = int(col.numel()); vals = torch.ones(nnz, dtype=torch.bool, device=dev)
A = gb.Matrix.from_csr(gb.BOOL, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
src = int(torch.argmax(rowptr[1:] - rowptr[:-1]))


def bfs(sync):
    rec = []
    def T(name, f):
        t = time.perf_counter(); r = f()
        if sync: torch.cuda.synchronize()
        rec.append((name, (time.perf_counter() - t) * 1e6)); return r
    t0 = time.perf_counter()
    v = gb.Vector.sparse(gb.UINT8, n); q = gb.Vector.sparse(gb.BOOL, n); q[src] = True
    level = 1
    while T("reduce_bool", q.reduce_bool) and level <= n:
        T("assign", lambda: v.assign_scalar(level, mask=q))
        T("vxm:" + str(level), lambda: v.vxm(A, mask=v, out=q, desc=D.RC))
        rec[-1] = (rec[-1][0] + ":" + gb.last_kernel_plan().split("<")[0], rec[-1][1])
        level += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6, rec

bfs(False); bfs(False)
for sync in ((False,) if args.only_async else (False, True)):
    best = None
    for _ in range(5):
        tot, rec = bfs(sync)
        if best is None or tot < best[0]: best = (tot, rec)
    print(f"--- {'synchronised after every call' if sync else 'as the loop runs'}: total {best[0]:.0f} us")
    for name, us in best[1]: print(f"   {name:40s} {us:8.1f} us")
