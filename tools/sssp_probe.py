#!/usr/bin/env python3
"""Where a sweep of the shortest-path loop goes: the reference's loop (v<accum MIN> = v MIN_PLUS A until nothing changes) on R-MAT-22
INT64 with a device synchronize after every mirror call.  usage: python tools/sssp_probe.py [--scale 22]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat

ap = argparse.ArgumentParser(); ap.add_argument("--scale", type=int, default=22); ap.add_argument("--only-async", action="store_true"); args = ap.parse_args()
S = args.scale; n = 1 << S; dev = torch.device("cuda", 0)
rowptr, col = rmat.csr_torch(S, dev, seed=42, drop_self_loops=True)
nnz = int(col.numel()); g = torch.Generator(device="cpu"); g.manual_seed(5)
vals = torch.randint(1, 256, (nnz,), generator=g, dtype=torch.int64).to(dev)
A = gb.Matrix.from_csr(gb.INT64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
src = int(torch.argmax(rowptr[1:] - rowptr[:-1]))


def run(sync):
    rec = []
    def T(name, f):
        t = time.perf_counter(); r = f()
        if sync: torch.cuda.synchronize()
        rec.append((name, (time.perf_counter() - t) * 1e6)); return r
    t0 = time.perf_counter()
    v = gb.Vector.sparse(gb.INT64, n); v[src] = 0
    rec.append(("(new vector, v[src] = 0)", (time.perf_counter() - t0) * 1e6))
    sweeps = 0
    while True:
        w = T("dup", v.dup)
        T("vxm:" + str(sweeps), lambda: v.vxm(A, semiring=gb.INT64.MIN_PLUS, accum=gb.INT64.MIN, out=v))
        rec[-1] = (rec[-1][0] + ":" + gb.last_kernel_plan().split("<")[0], rec[-1][1])
        sweeps += 1
        if T("iseq", lambda: w.iseq(v)): break
    t1 = time.perf_counter(); torch.cuda.synchronize()
    rec.append(("(final synchronise)", (time.perf_counter() - t1) * 1e6))
    rec.append(("(sum of the calls above)", sum(r[1] for r in rec)))
    return (time.perf_counter() - t0) * 1e6, rec

run(False); run(False)
for sync in ((False,) if args.only_async else (False, True)):
    best = min((run(sync) for _ in range(3)), key=lambda x: x[0])
    print(f"--- {'synchronised after every call' if sync else 'as the loop runs'}: total {best[0]:.0f} us")
    for name, us in best[1][:10] + best[1][-8:]: print(f"   {name:40s} {us:8.1f} us")
