#!/usr/bin/env python3
"""bench.py — the BASELINE.json metric on MI355X: GFLOP/s + GB/s (vs the HBM roofline) of the
GrB_mxv hot path on synthetic R-MAT.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input: one FP64 PLUS_TIMES
`A.mxv(x)` (GrB_mxv through the C ABI) with every operand already resident in HBM.
  N = 1 : BASELINE.json configs[1] — R-MAT scale-22 (n = 4 194 304, 16·2^22 sampled edges).
  N > 1 : weak scaling — R-MAT scale 22+log2(N) row-partitioned into N entry-balanced blocks
          (N = 8 is the scale-25 partition of configs[4]); each step every rank first receives the
          other ranks' slices of x (allgatherv: grouped RCCL send/recv over xGMI) and then multiplies
          its row block.  value = 2·(entries of all ranks)·K / max-over-ranks time.
Rank 0 prints ONE JSON line; see DESIGN.md §6 for how `roofline` and `cpu_baseline` are measured.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scale", type=int, default=22, help="R-MAT scale per GPU (22 = the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
            sys.exit(2)
    # test hook: BENCH_DEVICE_OVERRIDE puts every rank on one GPU (with BENCH_BACKEND=gloo) to exercise the N>1 code path on a 1-GPU box
    if "BENCH_DEVICE_OVERRIDE" in os.environ:
        local_rank = int(os.environ["BENCH_DEVICE_OVERRIDE"])
    os.environ["GRB_MI355X_DEVICE"] = str(local_rank)

    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import pygraphblas_amd as gb
    from pygraphblas_amd import rmat
    from pygraphblas_amd import dist as gdist
    lib = gb.lib
    info = gb.device_info()
    if not info["ok"]:
        print("bench.py: no HIP device: " + info["name"], file=sys.stderr)
        sys.exit(3)

    # ---- synthetic workload, generated in HBM -----------------------------------------------------
    log2w = world.bit_length() - 1
    assert 1 << log2w == world, "--gpus must be a power of two"
    scale = args.scale + log2w
    n = 1 << scale
    if world > 1:
        bounds = gdist.balanced_row_blocks(gdist.rmat_expected_row_prefix(scale), world)
    else:
        bounds = [0, n]
    r0, r1 = bounds[rank], bounds[rank + 1]
    t_gen = time.time()
    rowptr, col = rmat.csr_torch(scale, dev, seed=42, row_range=(r0, r1) if world > 1 else None)
    nnz = int(col.numel())
    vals = rmat.values_torch(nnz, dev, seed=43 + rank)
    x_all = rmat.values_torch(n, dev, seed=44)
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    A = gb.Matrix.from_csr(gb.FP64, r1 - r0, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    x = gb.Vector.from_dense_array((x_all.data_ptr(), n), gb.FP64, device=True)
    w = gb.Vector.sparse(gb.FP64, r1 - r0)
    del rowptr, col, vals
    torch.cuda.empty_cache()
    sr = gb.FP64.PLUS_TIMES
    xv_ptr, _, _ = x.device_view()
    x_view = gdist.as_torch(xv_ptr, n, "<f8", dev)          # the HBM buffer the kernel gathers from
    x_mine = x_all[r0:r1].clone()
    del x_all

    def step():
        if world > 1:
            gdist.allgatherv_into(x_view, x_mine, bounds, rank, world, dist)
        A.mxv(x, semiring=sr, out=w)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    plan = gb.last_kernel_plan()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    tot = torch.tensor([elapsed, float(nnz)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, nnz_total = float(mx[0]), int(sm[1])
    else:
        nnz_total = nnz
    ms_per_step = elapsed / args.steps * 1e3
    gflops = 2.0 * nnz_total * args.steps / elapsed / 1e9

    # ---- roofline of the dominant kernel: HIP events on the library's stream around K launches -------
    # algorithmic bytes per launch: nnz*(8+4) + (nrows+1)*4 + ncols*8 + nrows*8
    alg_bytes = nnz * 12 + (r1 - r0 + 1) * 4 + n * 8 + (r1 - r0) * 8
    torch.cuda.synchronize()
    lib.GrBX_timer_start()
    for _ in range(args.steps):
        A.mxv(x, semiring=sr, out=w)
    ms = C.c_float(0)
    lib.GrBX_timer_stop(C.byref(ms))
    kernel_ms = ms.value / args.steps
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "spmv_pmc_traffic.json")
    if os.path.exists(pmc_file) and world == 1 and args.scale == 22:     # the PMC passes were collected on this exact workload
        try:
            traffic = json.load(open(pmc_file)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                "traffic": traffic, "kernel": plan, "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes": alg_bytes,
                "frac_of_measured_copy_peak_6290": round(achieved / 6290.0, 4)}

    out = {
        "metric": "GFLOPS + GB/s (vs roofline) for mxv/mxm on R-MAT-22, 1/2/4/8 GPU",
        "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"R-MAT scale-{scale} FP64 PLUS_TIMES SpMV (GrB_mxv), edgefactor 16, "
                               f"{'row-partitioned into %d entry-balanced blocks + allgatherv of x' % world if world > 1 else 'BASELINE.json configs[1]'}",
                   "n": n, "nnz": nnz_total, "semiring": "PLUS_TIMES_FP64", "parallelism": f"rowblock{world}",
                   "graph_build_s": round(t_gen, 2), "device": info["name"]},
        "gbps_algorithmic": round(alg_bytes * args.steps / (elapsed if world == 1 else kernel_ms * 1e-3 * args.steps) / 1e9, 1),
        "roofline": roofline,
    }

    # ---- CPU baseline (rank 0, N=1 only): the oracle's typed OpenMP loop on a bounded sample ------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        rp, ci, av = A.to_csr()
        xh, _ = x.to_dense_arrays()
        y, pres = O.fast_spmv(rp, ci, av, xh)               # first call: page-in + parity check
        gy, gp = w.to_dense_arrays()
        ok = bool(np.array_equal(pres, gp) and np.allclose(gy[gp != 0], y[pres != 0], rtol=1e-6, atol=0.0))
        t1 = time.perf_counter(); O.fast_spmv(rp, ci, av, xh); one = time.perf_counter() - t1
        reps = max(3, min(100, int(10.0 / max(one, 1e-3))))
        t1 = time.perf_counter()
        for _ in range(reps):
            O.fast_spmv(rp, ci, av, xh)
        cpu_t = (time.perf_counter() - t1) / reps
        out["cpu_baseline"] = {"value": round(2.0 * nnz / cpu_t / 1e9, 3), "unit": "GFLOP/s", "cores": O.num_threads(), "kind": "port",
                               "sample": f"{reps} passes of the same scale-{scale} FP64 SpMV with oracle/grb_oracle.c fast_spmv_plus_times_fp64 "
                                         f"(OpenMP, {O.num_threads()} threads); SuiteSparse:GraphBLAS itself is not installed on this machine",
                               "ms_per_pass": round(cpu_t * 1e3, 2), "gbps_algorithmic": round(alg_bytes / cpu_t / 1e9, 2)}
        out["parity_vs_oracle"] = "ok (pattern exact, values rtol 1e-6)" if ok else "MISMATCH"
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
