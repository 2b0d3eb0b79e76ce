#!/usr/bin/env python3
"""Experiment harness (not part of the product): kernel X with S sub-panels per XCD (GRB_MI355X_XS), round 4.

  part A  R-MAT-`--scale` FP64 PLUS_TIMES mxv and FP32 PLUS_SECOND mxv for every S: time per product, plan build time, sub-rows, and the
          result against S = 1 (rtol 1e-9: the sub-row partition changes the association of the sums) and once against the oracle
  part B  R-MAT-`--pr-scale` FP32 PageRank (gap/prmark.py loop) for every S: ms per iteration, rank vector against S = 1
Prints one JSON line per measurement."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat, loops

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22)
ap.add_argument("--pr-scale", type=int, default=25)
ap.add_argument("--subpanels", default="1,2,4")
ap.add_argument("--pr-subpanels", default="1,4,8")
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--oracle", action="store_true")
ap.add_argument("--skip-a", action="store_true")
ap.add_argument("--skip-b", action="store_true")
ap.add_argument("--permute", action="store_true", help="part A on the label-permuted graph as well")
args = ap.parse_args()
dev = torch.device("cuda", 0)
lib = gb.lib


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.GrBX_timer_start()
    for _ in range(reps):
        fn()
    ms = C.c_float(0); lib.GrBX_timer_stop(C.byref(ms))
    return ms.value / reps


def plan_ms():
    ms = C.c_float(0); lib.GrBX_last_plan_build_ms(C.byref(ms)); return round(ms.value, 3)


def part_a(scale, permute_seed=None):
    n = 1 << scale
    kw = {} if permute_seed is None else {"permute_seed": permute_seed}
    rowptr, col = rmat.csr_torch(scale, dev, seed=42, **kw)
    nnz = int(col.numel())
    for tname, srname, ts in (("FP64", "PLUS_TIMES", 8), ("FP32", "PLUS_SECOND", 4)):
        T = getattr(gb, tname); tdt = torch.float64 if ts == 8 else torch.float32
        vals = rmat.values_torch(nnz, dev, seed=43, dtype=tdt)
        xs = rmat.values_torch(n, dev, seed=44, dtype=tdt)
        x = gb.Vector.from_dense_array((xs.data_ptr(), n), T, device=True)
        base = None
        for S in args.subpanels.split(","):
            if S == "a":                                                         # "a": the library's own choice
                os.environ.pop("GRB_MI355X_XOWN", None); os.environ.pop("GRB_MI355X_XS", None); own = None
            else:
                os.environ["GRB_MI355X_XOWN"] = "1" if S.endswith("o") else "0"      # "4o": four sub-panels per XCD, each with its own LDS table
                own = S.endswith("o"); S = int(S.rstrip("o"))
                os.environ["GRB_MI355X_XS"] = str(S)
            A = gb.Matrix.from_csr(T, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
            w = gb.Vector.sparse(T, n)
            sr = getattr(T, srname)
            t = timed(lambda: A.mxv(x, semiring=sr, out=w), args.reps)
            plan = gb.last_kernel_plan()
            y, pres = w.to_dense_arrays()
            rec = {"part": "A", "scale": scale, "permuted": permute_seed is not None, "type": tname, "semiring": srname, "S": S, "own_tables": own, "ms": round(t, 4), "plan_build_ms": plan_ms(), "plan": plan}
            uses_vals = srname == "PLUS_TIMES"
            alg = nnz * ((ts if uses_vals else 0) + 4) + (n + 1) * 4 + 2 * n * ts
            rec["alg_GBps"] = round(alg / t / 1e6, 1); rec["frac_of_8TBps"] = round(alg / t / 1e6 / 8000, 4)
            if base is None:
                base = (y, pres)
                if args.oracle and uses_vals and permute_seed is None:
                    from oracle import oracle as O
                    rp, ci, av = A.to_csr(); xh, _ = x.to_dense_arrays()
                    oy, op = O.fast_spmv(rp, ci, av, xh)
                    rec["vs_oracle"] = bool(np.array_equal(op, pres) and np.allclose(y[pres != 0], oy[op != 0], rtol=1e-6, atol=0))
            else:
                rec["vs_S1"] = bool(np.array_equal(pres, base[1]) and np.allclose(y[pres != 0], base[0][base[1] != 0], rtol=1e-9 if ts == 8 else 1e-5, atol=0))
            print(json.dumps(rec), flush=True)
            del A, w
    del rowptr, col


def part_b(scale):
    n = 1 << scale
    t0 = time.perf_counter()
    rowptr, col = rmat.csr_torch(scale, dev, seed=42)
    nnz = int(col.numel())
    ones = torch.ones(nnz, dtype=torch.float32, device=dev)
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.float32)
    pres = (deg > 0).to(torch.uint8)
    torch.cuda.synchronize()
    print(json.dumps({"part": "B", "scale": scale, "nnz": nnz, "graph_build_s": round(time.perf_counter() - t0, 2)}), flush=True)
    base = None
    for S in args.pr_subpanels.split(","):
        if S == "a":
            os.environ.pop("GRB_MI355X_XOWN", None); os.environ.pop("GRB_MI355X_XS", None); own = None
        else:
            os.environ["GRB_MI355X_XOWN"] = "1" if S.endswith("o") else "0"
            own = S.endswith("o"); S = int(S.rstrip("o"))
            os.environ["GRB_MI355X_XS"] = str(S)
        A = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (ones.data_ptr(), nnz), device=True)

        def degrees():
            return gb.Vector.from_dense_array((deg.data_ptr(), n), gb.FP32, present=pres.data_ptr(), device=True)
        t1 = time.perf_counter(); loops.pagerank(A, degrees(), fixed_iterations=2); torch.cuda.synchronize(); first = time.perf_counter() - t1
        pm = plan_ms()
        times = []
        for _ in range(3):
            torch.cuda.synchronize(); t1 = time.perf_counter()
            r, its, rdiff = loops.pagerank(A, degrees(), fixed_iterations=10)
            torch.cuda.synchronize(); times.append((time.perf_counter() - t1) / its)
        ms = sorted(times)[1] * 1e3
        alg = nnz * 4 + (n + 1) * 4 + n * 4 + n * 4 + 6 * n * 4
        rv = r.to_dense_arrays()[0]
        rec = {"part": "B", "scale": scale, "S": S, "own_tables": own, "ms_per_iteration": round(ms, 4), "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4), "first_two_iterations_s": round(first, 3), "plan_build_ms": pm,
               "plan": gb.last_kernel_plan()}
        if base is None:
            base = rv
        else:
            rec["vs_S1"] = bool(np.allclose(rv, base, rtol=1e-5, atol=0))
        print(json.dumps(rec), flush=True)
        del A, r


if not args.skip_a:
    part_a(args.scale)
    if args.permute:
        part_a(args.scale, permute_seed=7)
if not args.skip_b:
    part_b(args.pr_scale)
