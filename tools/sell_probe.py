#!/usr/bin/env python3
"""Experiment harness (not part of the product): kernel X's lane-per-piece layout (grb_spmv_sell.hpp, round 6) against the tile pipeline.

  part A  R-MAT-`--scale` mxv for (type, semiring) pairs: per variant — tiles (GRB_MI355X_SELL=0), lane-per-piece (=1), and the tile pipeline on
          the plan that carries both layouts (its sub-rows cut into pieces: GRB_MI355X_SELL=1, GRB_MI355X_SELL_RUN=0) — the time per product,
          the plan build time, the result against the tile pipeline's (rtol 1e-9 / 1e-5: the partition of the sums differs; integers exact),
          two runs bit for bit, and with --oracle the FP64 result against the CPU oracle
  part B  R-MAT-`--pr-scale` FP32 PageRank (gap/prmark.py loop): ms per iteration for both layouts, rank vector against the tile pipeline's
Prints one JSON line per measurement."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat, loops

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22)
ap.add_argument("--pr-scale", type=int, default=22)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--oracle", action="store_true")
ap.add_argument("--skip-a", action="store_true")
ap.add_argument("--skip-b", action="store_true")
ap.add_argument("--cases", default="FP64:PLUS_TIMES,FP32:PLUS_SECOND,INT64:MIN_PLUS,FP32:PLUS_TIMES")
ap.add_argument("--variants", default="tiles,sell,tiles_on_pieces")
args = ap.parse_args()
dev = torch.device("cuda", 0)
lib = gb.lib
VARIANTS = {"tiles": ("0", "1"), "sell": ("1", "1"), "tiles_on_pieces": ("1", "0")}


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


def part_a(scale):
    n = 1 << scale
    rowptr, col = rmat.csr_torch(scale, dev, seed=42)
    nnz = int(col.numel())
    for case in args.cases.split(","):
        tname, srname = case.split(":")
        T = getattr(gb, tname)
        tdt = {"FP64": torch.float64, "FP32": torch.float32, "INT64": torch.int64, "INT32": torch.int32}[tname]
        ts = torch.empty(0, dtype=tdt).element_size()
        if tdt.is_floating_point:
            vals = rmat.values_torch(nnz, dev, seed=43, dtype=tdt); xs = rmat.values_torch(n, dev, seed=44, dtype=tdt)
        else:
            vals = (rmat.values_torch(nnz, dev, seed=43, dtype=torch.float64) * 1000).to(tdt); xs = (rmat.values_torch(n, dev, seed=44, dtype=torch.float64) * 1000).to(tdt)
        x = gb.Vector.from_dense_array((xs.data_ptr(), n), T, device=True)
        base = None
        for var in args.variants.split(","):
            os.environ["GRB_MI355X_SELL"], os.environ["GRB_MI355X_SELL_RUN"] = VARIANTS[var]
            A = gb.Matrix.from_csr(T, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
            w = gb.Vector.sparse(T, n)
            sr = getattr(T, srname)
            os.environ["GRB_MI355X_XPLAN_AFTER"] = "0"
            t = timed(lambda: A.mxv(x, semiring=sr, out=w), args.reps)
            plan = gb.last_kernel_plan()
            y, pres = w.to_dense_arrays()
            A.mxv(x, semiring=sr, out=w)
            y2, pres2 = w.to_dense_arrays()
            uses_vals = srname != "PLUS_SECOND"
            alg = nnz * ((ts if uses_vals else 0) + 4) + (n + 1) * 4 + 2 * n * ts
            rec = {"part": "A", "scale": scale, "type": tname, "semiring": srname, "variant": var, "ms": round(t, 4), "alg_GBps": round(alg / t / 1e6, 1), "frac_of_8TBps": round(alg / t / 1e6 / 8000, 4),
                   "plan_build_ms": plan_ms(), "plan": plan, "same_bits_twice": bool(np.array_equal(y.view(np.uint8), y2.view(np.uint8)) and np.array_equal(pres, pres2))}
            if base is None:
                base = (y, pres)
                if args.oracle and tname == "FP64" and srname == "PLUS_TIMES":
                    from oracle import oracle as O
                    rp, ci, av = A.to_csr(); xh, _ = x.to_dense_arrays()
                    oy, op = O.fast_spmv(rp, ci, av, xh)
                    base = (oy, op); rec["vs_oracle"] = bool(np.array_equal(op, pres) and np.allclose(y[pres != 0], oy[op != 0], rtol=1e-6, atol=0))
            else:
                m = pres != 0
                same_pattern = bool(np.array_equal(pres, base[1]))
                if tdt.is_floating_point:
                    ok = same_pattern and bool(np.allclose(y[m], base[0][m], rtol=1e-9 if ts == 8 else 1e-5, atol=0))
                    if same_pattern and m.any():
                        rec["max_rel"] = float(np.max(np.abs(y[m] - base[0][m]) / np.maximum(np.abs(base[0][m]), 1e-300)))
                else:
                    ok = same_pattern and bool(np.array_equal(y[m], base[0][m]))
                rec["vs_first_variant"] = ok
                if not ok:
                    bad = np.nonzero((pres != base[1]) | ((pres != 0) & (y != base[0])))[0]
                    rec["mismatches"] = int(bad.size); rec["first_bad"] = [int(b) for b in bad[:8]]
            print(json.dumps(rec), flush=True)
            del A, w
    del rowptr, col


def part_b(scale):
    n = 1 << scale
    rowptr, col = rmat.csr_torch(scale, dev, seed=42)
    nnz = int(col.numel())
    ones = torch.ones(nnz, dtype=torch.float32, device=dev)
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.float32)
    pres = (deg > 0).to(torch.uint8)
    torch.cuda.synchronize()
    base = None
    for var in args.variants.split(","):
        if var == "tiles_on_pieces":
            continue
        os.environ["GRB_MI355X_SELL"], os.environ["GRB_MI355X_SELL_RUN"] = VARIANTS[var]
        os.environ.pop("GRB_MI355X_XPLAN_AFTER", None)
        A = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (ones.data_ptr(), nnz), device=True)

        def degrees():
            return gb.Vector.from_dense_array((deg.data_ptr(), n), gb.FP32, present=pres.data_ptr(), device=True)
        t1 = time.perf_counter(); loops.pagerank(A, degrees(), fixed_iterations=2); torch.cuda.synchronize(); first = time.perf_counter() - t1
        pm = plan_ms()
        times = []
        for _ in range(5):
            torch.cuda.synchronize(); t1 = time.perf_counter()
            r, its, rdiff = loops.pagerank(A, degrees(), fixed_iterations=20)
            torch.cuda.synchronize(); times.append((time.perf_counter() - t1) / its)
        ms = sorted(times)[2] * 1e3
        alg = nnz * 4 + (n + 1) * 4 + n * 4 + n * 4 + 6 * n * 4
        rv = r.to_dense_arrays()[0]
        rec = {"part": "B", "scale": scale, "variant": var, "ms_per_iteration": round(ms, 4), "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4), "first_two_iterations_s": round(first, 3), "plan_build_ms": pm,
               "plan": gb.last_kernel_plan()}
        if base is None:
            base = rv
        else:
            rec["vs_first_variant"] = bool(np.allclose(rv, base, rtol=1e-5, atol=0))
        print(json.dumps(rec), flush=True)
        del A, r


if not args.skip_a:
    part_a(args.scale)
if not args.skip_b:
    part_b(args.pr_scale)
