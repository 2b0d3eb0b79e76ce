#!/usr/bin/env python3
"""bench.py — the BASELINE.json metric on MI355X: GFLOP/s + GB/s (vs the HBM roofline) of the
GrB_mxv / GrB_mxm hot path on synthetic R-MAT.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input: one FP64 PLUS_TIMES `A.mxv(x)` (GrB_mxv through
the C ABI) with every operand already resident in HBM — `value`, `ms_per_step` and `roofline` are about that step.
  N = 1 : BASELINE.json configs[1] — R-MAT scale-22 (n = 4 194 304, 16·2^22 sampled edges).
  N > 1 : `--scaling weak` (default): R-MAT scale 22+log2(N) row-partitioned into N entry-balanced blocks (N = 8 is the
          scale-25 partition of configs[4]); `--scaling strong`: the fixed R-MAT-`--scale` graph over N ranks.  Each step the
          ranks' slices of x are exchanged by the library's allgatherv (RCCL send/recv over xGMI on its own stream,
          pygraphblas_amd/csrc/grb_dist.cpp) while the diagonal block of the row block is multiplied, then the off-diagonal
          block.  value = 2·(entries of all ranks)·K / max-over-ranks time.  The other mode's SpMV is reported beside it
          (`spmv_strong` / `spmv_weak`) with per-phase times (exchange, diagonal block, off-diagonal block).
The timed region is run `--blocks` times (default 5), each block EXACTLY K steps between barrier + synchronize; the line
reports the median block (`ms_per_step_blocks` lists all of them).
Beside the step the JSON line carries, every one with its own `roofline`:
  spmv_extra        what the plan of the SpMV kernel costs, the rate without a plan, the rate on permuted labels   [N = 1]
  mxm               configs[3]: triangle count L.mxm(L, PLUS_PAIR, mask=L).reduce_int() on R-MAT-22, bit-exact; for N > 1
                    the rows of L and of the mask in flop-balanced blocks, L replicated, INT64 all-reduce          [every N]
  bfs               configs[2]: the reference's BOOL LOR_LAND BFS loop on R-MAT-22, bit-exact level vector; for N > 1 the
                    frontier is gathered as bits                                                                   [every N]
  pagerank          configs[4]'s loop (gap/prmark.py, FP32 PLUS_SECOND) on the step's partition                    [every N]
  pagerank_scale25  configs[4] at its stated size: R-MAT scale-25 on one GPU / over the N ranks                    [every N]
  sssp              the reference's MIN_PLUS shortest-path loop on R-MAT-22, INT64 weights, bit-exact              [N = 1]
  aa                the unmasked A @ A (GrB_mxm, mask = NULL) on R-MAT-18 FP64: the two-pass LDS-hash Gustavson, sampled rows of
                    the result against the oracle                                                                  [N = 1]
  bc                gap/bcmark.py's batched betweenness centrality (ns = 4) on R-MAT-22, against the oracle        [N = 1]
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

PEAK = 8000.0          # GB/s, MI355X HBM3E (spec); 6290 GB/s is the measured copy rate of /opt/skills/guides/MI355X_MICROARCH.md


def roof(alg_bytes, seconds, **extra):
    a = alg_bytes / seconds / 1e9
    r = {"bound": "hbm", "achieved": round(a, 1), "peak": PEAK, "unit": "GB/s", "frac": round(a / PEAK, 4), "algorithmic_bytes": int(alg_bytes),
         "frac_of_measured_copy_peak_6290": round(a / 6290.0, 4)}
    r.update(extra)
    return r


def first_run(cx, fn):
    """(result, seconds) of the FIRST call of a workload on its freshly built operands — what a caller who runs it once pays (SURVEY.md 8d:
    "additionally report end-to-end call time through the C ABI"): the cached transpose, row heads, kernel plans, row-binning buffers and the
    pool's first allocations are all inside.  Device work is complete when the clock stops."""
    cx.torch.cuda.synchronize(); cx.barrier(); t = time.perf_counter()
    res = fn()
    cx.torch.cuda.synchronize(); cx.barrier()
    return res, cx.max_over_ranks(time.perf_counter() - t)


def static_traffic(name):
    """HBM-side bytes per launch from the PMC passes of an earlier, separate run of the same workload (profiles/*.json)."""
    f = os.path.join(ROOT, "profiles", name)
    try:
        d = json.load(open(f))
        return d.get("hbm_bytes_per_launch"), f"profiles/{name} (static: rocprofv3 --pmc passes of this workload in their own runs, {d.get('collected_with', 'tools/pmc_*.sh')}; not measured in this run)"
    except Exception:       # noqa: BLE001
        return None, None


class Ctx:
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of exactly --steps steps; the median block is reported")
    ap.add_argument("--scale", type=int, default=22, help="R-MAT scale (22 = the BASELINE config): per GPU for weak scaling, of the whole graph for strong")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="what the headline step does for N > 1")
    ap.add_argument("--pr-scale", type=int, default=25, help="scale of the pagerank_scale25 sub-object (configs[4])")
    ap.add_argument("--aa-scale", type=int, default=18, help="scale of the unmasked A @ A sub-object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed step (no sub-objects)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
            sys.exit(2)
    # test hook: BENCH_DEVICE_OVERRIDE puts every rank on one GPU (with BENCH_TRANSPORT=host) to exercise the N>1 code path on a 1-GPU box
    if "BENCH_DEVICE_OVERRIDE" in os.environ:
        local_rank = int(os.environ["BENCH_DEVICE_OVERRIDE"])
    os.environ["GRB_MI355X_DEVICE"] = str(local_rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # (the host driver supports dmabuf IPC only: RCCL between processes needs this — normally exported already)

    import numpy as np
    import torch
    import torch.distributed as tdist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # torch.distributed is the control plane only (the 128-byte communicator id, barriers, the max over ranks of the
        # timings): gloo on the host.  Every byte of the data path moves through the library's own RCCL communicator.
        tdist.init_process_group("gloo", rank=rank, world_size=world)

    import pygraphblas_amd as gb
    from pygraphblas_amd import rmat, loops
    from pygraphblas_amd import dist as gdist
    lib = gb.lib
    info = gb.device_info()
    if not info["ok"]:
        print("bench.py: no HIP device: " + info["name"], file=sys.stderr)
        sys.exit(3)

    def share(ident):
        box = [ident]
        tdist.broadcast_object_list(box, src=0)
        return box[0]
    transport = os.environ.get("BENCH_TRANSPORT", "rccl")
    transport_note = transport
    if world > 1 and transport == "rccl":
        # the data path is the library's RCCL communicator.  It is set up and self-tested (an all-reduce of rank + 1) on every
        # rank; if any rank fails, ALL ranks fall back to host copies over gloo (slower, and labelled so in the line) instead
        # of leaving the driver without a number.
        ok, why = 1, ""
        try:
            comm = gdist.Comm(rank, world, transport="rccl", share=share, tdist=tdist)
            ok = int(comm.allreduce(rank + 1, gb.INT64, "PLUS") == world * (world + 1) // 2)
            why = "" if ok else "all-reduce self-test returned a wrong sum"
        except Exception as e:          # noqa: BLE001 - any failure of the communicator selects the fallback
            ok, why = 0, f"{type(e).__name__}: {e}"
        flag = torch.tensor([ok], dtype=torch.int64); tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
        if int(flag[0]) == 0:
            try:
                comm.close()
            except Exception:           # noqa: BLE001
                pass
            print(f"bench.py[{rank}]: RCCL data path unavailable ({why or 'another rank failed'}); falling back to host transport", file=sys.stderr)
            comm = gdist.Comm(rank, world, transport="host", share=share, tdist=tdist)
            transport_note = "host copies over gloo (RCCL set-up failed on some rank; NOT the designed data path: read this line as UNMEASURED)"
        elif gdist.bound_transport().endswith("libfake_rccl.so"):
            transport_note = ("rccl-abi (fake, hipIpc): the library's exchange path (grouped ncclSend/ncclRecv, second stream, ncclAllReduce) between ranks that "
                              "SHARE ONE GPU, through tests/libfake_rccl.so — a test stand-in, host-synchronous, no xGMI: NOT a scaling number")
    else:
        comm = gdist.Comm(rank, world, transport=transport, share=share, tdist=tdist if world > 1 else None)
        if world > 1 and transport == "host":
            transport_note = "host copies over gloo (test transport: NOT the designed data path)"

    cx = Ctx()
    cx.gb, cx.rmat, cx.loops, cx.gdist, cx.torch, cx.np, cx.lib, cx.dev, cx.comm = gb, rmat, loops, gdist, torch, np, lib, dev, comm
    cx.rank, cx.world, cx.tdist, cx.args = rank, world, tdist, args

    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64); tdist.all_reduce(t, op=tdist.ReduceOp.MAX); return float(t[0])

    def sum_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64); tdist.all_reduce(t, op=tdist.ReduceOp.SUM); return float(t[0])
    cx.barrier, cx.max_over_ranks, cx.sum_over_ranks = barrier, max_over_ranks, sum_over_ranks

    log2w = world.bit_length() - 1
    assert 1 << log2w == world, "--gpus must be a power of two"
    weak_scale, strong_scale = args.scale + log2w, args.scale
    head_scale = weak_scale if (args.scaling == "weak" or world == 1) else strong_scale
    head = SpmvJob(cx, head_scale)
    res = head.run(args.steps, args.warmup, args.blocks)
    n, nnz_total = head.n, res["nnz_total"]
    traffic, traffic_source = (None, None)
    if world == 1 and head_scale == 22:       # the PMC passes were collected on this exact workload, in their own runs
        traffic, traffic_source = static_traffic("spmv_pmc_traffic.json")
    roofline = roof(res["alg_bytes"], res["kernel_ms"] * 1e-3, traffic=traffic, traffic_source=traffic_source, kernel=res["plan"], kernel_ms=round(res["kernel_ms"], 4))
    out = {
        "metric": "GFLOPS + GB/s (vs roofline) for mxv/mxm on R-MAT-22, 1/2/4/8 GPU",
        "value": round(res["gflops"], 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(res["ms_per_step"], 4), "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"R-MAT scale-{head_scale} FP64 PLUS_TIMES SpMV (GrB_mxv), edgefactor 16, "
                               f"{'row-partitioned into %d entry-balanced blocks + allgatherv of x (RCCL, overlapped with the diagonal block)' % world if world > 1 else 'BASELINE.json configs[1]'}",
                   "n": n, "nnz": nnz_total, "semiring": "PLUS_TIMES_FP64", "parallelism": f"rowblock{world}", "transport": transport_note if world > 1 else "none (one GPU)",
                   "graph_build_s": round(head.t_gen, 2), "device": info["name"]},
        "ms_per_step_blocks": [round(b, 4) for b in res["blocks_ms"]],
        "timing": f"median of {len(res['blocks_ms'])} blocks of exactly {args.steps} steps, each between barrier + synchronize (max over ranks)",
        "gbps_algorithmic": round(res["alg_bytes_total"] / (res["ms_per_step"] * 1e-3) / 1e9, 1),
        "roofline": roofline,
    }
    if world > 1:
        out["phases"] = head.phases(min(args.steps, 20))

    # ---- CPU baseline (rank 0, N=1 only): the oracle's typed OpenMP loop on a bounded sample ------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        A, x, w = head.mats[0], head.x, head.w
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
        out["cpu_baseline"] = {"value": round(2.0 * head.nnz / cpu_t / 1e9, 3), "unit": "GFLOP/s", "cores": O.num_threads(), "kind": "port",
                               "sample": f"{reps} passes of the same scale-{head_scale} FP64 SpMV with oracle/grb_oracle.c fast_spmv_plus_times_fp64 "
                                         f"(OpenMP, {O.num_threads()} threads); SuiteSparse:GraphBLAS itself is not installed on this machine",
                               "ms_per_pass": round(cpu_t * 1e3, 2), "gbps_algorithmic": round(res["alg_bytes"] / cpu_t / 1e9, 2)}
        out["parity_vs_oracle"] = "ok (pattern exact, values rtol 1e-6)" if ok else "MISMATCH"
        del rp, ci, av, xh, y, pres

    if not args.no_extras:
        if world == 1:
            out["spmv_extra"] = spmv_extra(cx, head, res)
        bounds_head = head.bounds
        head.release(); del head
        torch.cuda.empty_cache()
        if world > 1:
            other_scale = strong_scale if args.scaling == "weak" else weak_scale
            other = SpmvJob(cx, other_scale)
            r2 = other.run(min(args.steps, 20), min(args.warmup, 3), 3)
            key = "spmv_strong" if args.scaling == "weak" else "spmv_weak"
            out[key] = {"workload": f"R-MAT scale-{other_scale} FP64 PLUS_TIMES SpMV over {world} ranks ({'fixed graph: strong scaling' if key == 'spmv_strong' else 'scale grows with N: weak scaling'})",
                        "n": other.n, "nnz": r2["nnz_total"], "ms_per_step": round(r2["ms_per_step"], 4), "GFLOPS": round(r2["gflops"], 1),
                        "roofline": roof(r2["alg_bytes"], r2["kernel_ms"] * 1e-3, kernel=r2["plan"], kernel_ms=round(r2["kernel_ms"], 4), note="this rank's row block"),
                        "phases": other.phases(min(args.steps, 20))}
            other.release(); del other
            torch.cuda.empty_cache()
        out["mxm"] = bench_triangles(cx, args.scale)
        out["bfs"] = bench_bfs(cx, args.scale)
        out["pagerank"] = bench_pagerank(cx, head_scale, bounds_head, min(args.steps, 20), "pagerank")
        if args.pr_scale > 0:
            pr_bounds = gdist.balanced_row_blocks(gdist.rmat_expected_row_prefix(args.pr_scale), world) if world > 1 else [0, 1 << args.pr_scale]
            out["pagerank_scale25"] = bench_pagerank(cx, args.pr_scale, pr_bounds, min(args.steps, 10), "pagerank_scale25")
        if world == 1:
            out["sssp"] = bench_sssp(cx, args.scale)
            torch.cuda.empty_cache()
            out["aa"] = bench_aa(cx, args.aa_scale)
            torch.cuda.empty_cache()
            out["aa_wide"] = bench_aa(cx, 20, edgefactor=4)       # 2^20 columns: four times what a row's bitmap + accumulators fit the LDS for (missing #3)
            torch.cuda.empty_cache()
            out["bc"] = bench_bc(cx, args.scale)
            out["mxm_fp64_deterministic"] = bench_masked_fp64_deterministic(cx, args.scale)

    if rank == 0:
        print(json.dumps(out), flush=True)
    comm.close()
    if world > 1:
        tdist.destroy_process_group()


class SpmvJob:
    """The timed step: FP64 PLUS_TIMES GrB_mxv on this rank's row block of R-MAT-`scale` (the whole graph for one rank)."""

    def __init__(self, cx, scale):
        gb, rmat, gdist, torch, dev = cx.gb, cx.rmat, cx.gdist, cx.torch, cx.dev
        self.cx, self.scale = cx, scale
        world, rank = cx.world, cx.rank
        n = self.n = 1 << scale
        self.bounds = gdist.balanced_row_blocks(gdist.rmat_expected_row_prefix(scale), world) if world > 1 else [0, n]
        r0, r1 = self.r0, self.r1 = self.bounds[rank], self.bounds[rank + 1]
        t_gen = time.time()
        rowptr, col = rmat.csr_torch(scale, dev, seed=42, row_range=(r0, r1) if world > 1 else None)
        nnz = self.nnz = int(col.numel())
        vals = rmat.values_torch(nnz, dev, seed=43 + rank)
        x_all = rmat.values_torch(n, dev, seed=44)
        torch.cuda.synchronize()
        self.t_gen = time.time() - t_gen
        self.mats = []
        if world > 1:       # diagonal block (the columns this rank owns) and the rest: the first needs no remote data
            for rp_, c_, v_ in gdist.split_csr_columns(rowptr, col, r0, r1, vals):
                self.mats.append(gb.Matrix.from_csr(gb.FP64, r1 - r0, n, rp_.data_ptr(), c_.data_ptr(), (v_.data_ptr(), int(c_.numel())), device=True))
            del rp_, c_, v_
        else:
            self.mats.append(gb.Matrix.from_csr(gb.FP64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True))
        self.x = gb.Vector.from_dense_array((x_all.data_ptr(), n), gb.FP64, device=True)           # the full operand (every slice valid at the start)
        self.x_mine = gb.Vector.from_dense_array((x_all[r0:r1].contiguous().data_ptr(), r1 - r0), gb.FP64, device=True)
        self.w = gb.Vector.sparse(gb.FP64, r1 - r0)
        del rowptr, col, vals, x_all
        torch.cuda.empty_cache()
        self.sr = gb.FP64.PLUS_TIMES
        # algorithmic bytes per launch: nnz*(8+4) + (nrows+1)*4 + ncols*8 + nrows*8
        self.alg_bytes = nnz * 12 + (r1 - r0 + 1) * 4 + n * 8 + (r1 - r0) * 8

    def step(self):
        cx = self.cx
        if cx.world > 1:
            cx.comm.allgatherv_start(self.x, self.x_mine, self.bounds)    # remote slices of x -> the HBM buffer the kernel gathers from
            self.mats[0].mxv(self.x, semiring=self.sr, out=self.w)        # diagonal block: local columns only
            cx.comm.wait()
            self.mats[1].mxv(self.x, semiring=self.sr, out=self.w, accum=cx.gb.FP64.PLUS)
        else:
            self.mats[0].mxv(self.x, semiring=self.sr, out=self.w)

    def run(self, steps, warmup, blocks):
        cx = self.cx
        lib, torch = cx.lib, cx.torch
        # the matrix's first product runs kernel W, the second builds kernel X's plan (the library's plan policy): both inside the warm-up
        # when it has >= 2 steps (the driver's runs: 5)
        for _ in range(min(2, max(1, warmup))):
            self.step()
        ms = C.c_float(0); lib.GrBX_last_plan_build_ms(C.byref(ms)); self.plan_build_ms = round(ms.value, 2)    # the second product built the plan of kernel X
        for _ in range(max(0, warmup - 2)):
            self.step()
        plan = cx.gb.last_kernel_plan()
        blocks_ms = []
        for _ in range(max(1, blocks)):
            cx.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            cx.barrier()
            blocks_ms.append(cx.max_over_ranks(time.perf_counter() - t0) / steps * 1e3)
        ms_per_step = sorted(blocks_ms)[len(blocks_ms) // 2]
        nnz_total = int(cx.sum_over_ranks(float(self.nnz)))
        # roofline of the dominant kernel(s): HIP events on the library's stream around K launches
        kms = []
        for _ in range(max(1, min(blocks, 3))):
            torch.cuda.synchronize()
            lib.GrBX_timer_start()
            for _ in range(steps):
                self.step()
            m = C.c_float(0); lib.GrBX_timer_stop(C.byref(m)); kms.append(m.value / steps)
        kernel_ms = sorted(kms)[len(kms) // 2]
        return {"ms_per_step": ms_per_step, "blocks_ms": blocks_ms, "gflops": 2.0 * nnz_total / (ms_per_step * 1e-3) / 1e9, "nnz_total": nnz_total, "kernel_ms": kernel_ms,
                "alg_bytes": self.alg_bytes, "alg_bytes_total": int(cx.sum_over_ranks(float(self.alg_bytes))), "plan": plan}

    def phases(self, reps):
        """Per-phase times of the N > 1 step on this rank, each measured alone (HIP events on the library's stream; the exchange
        by the host clock around start + wait + synchronize): what overlaps in the step is the exchange and the diagonal block."""
        cx = self.cx
        lib, torch = cx.lib, cx.torch

        def ev(fn):
            fn(); torch.cuda.synchronize(); lib.GrBX_timer_start()
            for _ in range(reps):
                fn()
            m = C.c_float(0); lib.GrBX_timer_stop(C.byref(m)); return m.value / reps
        diag = ev(lambda: self.mats[0].mxv(self.x, semiring=self.sr, out=self.w))
        off = ev(lambda: self.mats[1].mxv(self.x, semiring=self.sr, out=self.w, accum=cx.gb.FP64.PLUS))
        cx.barrier(); t = time.perf_counter()
        for _ in range(reps):
            cx.comm.allgatherv_start(self.x, self.x_mine, self.bounds); cx.comm.wait(); torch.cuda.synchronize()
        cx.barrier(); ex = (time.perf_counter() - t) / reps * 1e3
        return {"exchange_ms": round(cx.max_over_ranks(ex), 4), "diag_ms": round(cx.max_over_ranks(diag), 4), "offdiag_ms": round(cx.max_over_ranks(off), 4),
                "exchange_bytes_received_per_rank": int((self.n - (self.r1 - self.r0)) * 8), "note": "max over ranks; each phase timed alone"}

    def release(self):
        self.mats = []; self.x = self.x_mine = self.w = None


def spmv_extra(cx, head, res):
    """What the plan costs, and the rates without it / on permuted labels (N = 1)."""
    gb, rmat, torch, dev, lib = cx.gb, cx.rmat, cx.torch, cx.dev, cx.lib
    n, scale = head.n, head.scale

    def timed_mxv(Amat, xv, wv, reps):
        for _ in range(3):
            Amat.mxv(xv, semiring=head.sr, out=wv)
        torch.cuda.synchronize(); lib.GrBX_timer_start()
        for _ in range(reps):
            Amat.mxv(xv, semiring=head.sr, out=wv)
        m = C.c_float(0); lib.GrBX_timer_stop(C.byref(m)); return m.value / reps

    def plan_ms():
        ms = C.c_float(0); lib.GrBX_last_plan_build_ms(C.byref(ms)); return round(ms.value, 2)
    extra = {"plan_build_ms_first_in_process": head.plan_build_ms}      # includes the one-time loading of the ~40 plan-building kernels' code objects
    for key, env, what in (("rowblock", "adaptive", "k_spmv_adaptive (row-block kernel A: its row-block list is built in one host pass)"),
                           ("without_plan", "wavepipe", "k_spmv_wavepipe (kernel W: the pipeline on the matrix as stored — what a matrix runs BEFORE kernel X's panel-major plan exists, "
                                                        "i.e. its first product; W's own plan is a sampled column ranking + one re-labelling pass, no panel copy)")):
        os.environ["GRB_MI355X_SPMV"] = env
        try:
            t_a = timed_mxv(head.mats[0], head.x, head.w, 10)
            extra[f"ms_per_step_{key}"] = round(t_a, 4)
            extra[f"frac_{key}"] = round(res["alg_bytes"] / (t_a * 1e-3) / 1e9 / PEAK, 4)
            extra[f"{key}_kernel"] = what
        finally:
            os.environ.pop("GRB_MI355X_SPMV")
    head.release()
    torch.cuda.empty_cache()
    rp2, c2 = rmat.csr_torch(scale, dev, seed=42, permute_seed=7)
    v2 = rmat.values_torch(int(c2.numel()), dev, seed=43)
    x2t = rmat.values_torch(n, dev, seed=44)
    A2 = gb.Matrix.from_csr(gb.FP64, n, n, rp2.data_ptr(), c2.data_ptr(), (v2.data_ptr(), int(c2.numel())), device=True)
    x2 = gb.Vector.from_dense_array((x2t.data_ptr(), n), gb.FP64, device=True)
    w2 = gb.Vector.sparse(gb.FP64, n)
    # what the FIRST product of a fresh matrix costs end to end (host clock, synchronised): kernel W + its plan; the second builds kernel X's plan
    torch.cuda.synchronize(); t0 = time.perf_counter()
    A2.mxv(x2, semiring=head.sr, out=w2); lib.GrBX_device_synchronize()
    extra["first_call_ms"] = round((time.perf_counter() - t0) * 1e3, 3); extra["first_call_kernel"] = gb.last_kernel_plan()
    t0 = time.perf_counter()
    A2.mxv(x2, semiring=head.sr, out=w2); lib.GrBX_device_synchronize()
    extra["second_call_ms"] = round((time.perf_counter() - t0) * 1e3, 3); extra["second_call_kernel"] = gb.last_kernel_plan()
    extra["plan_policy"] = "first full-operand product of a matrix: kernel W (plan < 1 ms); the second builds kernel X's plan (GRB_MI355X_XPLAN_AFTER=1)"
    t_p = timed_mxv(A2, x2, w2, 20)
    alg2 = int(c2.numel()) * 12 + (n + 1) * 4 + 2 * n * 8
    warm = plan_ms()                                                # a second matrix of the same size in the same process: what a plan costs
    extra["plan_build_ms"] = warm
    extra["plan_build_in_steps"] = round(warm / res["ms_per_step"], 1)
    extra["permuted_labels"] = {"ms_per_step": round(t_p, 4), "GFLOPS": round(2.0 * int(c2.numel()) / t_p / 1e6, 1),
                                "frac": round(alg2 / (t_p * 1e-3) / 1e9 / PEAK, 4), "plan_build_ms": warm, "kernel": gb.last_kernel_plan(),
                                "note": "same R-MAT-22 with vertex labels permuted pseudo-randomly (rmat.py permute_seed=7)"}
    return extra


def _rows_slice(cx, typ, rowptr, col, vals, r0, r1, ncols):
    """Rows [r0, r1) of a CSR held as torch tensors, as a library matrix."""
    torch = cx.torch
    rp = (rowptr[r0:r1 + 1].to(torch.int64) & 0xFFFFFFFF)
    b, e = int(rp[0]), int(rp[-1])
    rp32 = (rp - b).to(torch.int32).contiguous()
    c = col[b:e].contiguous(); v = vals[b:e].contiguous()
    return cx.gb.Matrix.from_csr(typ, r1 - r0, ncols, rp32.data_ptr(), c.data_ptr(), (v.data_ptr(), e - b), device=True), e - b


def bench_triangles(cx, scale):
    """configs[3]: L.mxm(L, PLUS_PAIR, mask=L).reduce_int() on R-MAT-`scale`, checked against the oracle's count.  N > 1: the rows
    of A and of the mask in flop-balanced blocks, L replicated, the INT64 counts all-reduced (pygraphblas_amd.dist.triangle_count)."""
    gb, rmat, torch, dev, np, world, rank = cx.gb, cx.rmat, cx.torch, cx.dev, cx.np, cx.world, cx.rank
    n = 1 << scale
    rowptr, col = rmat.csr_torch(scale, dev, seed=42, symmetric=True, drop_self_loops=True, lower=True)
    nnz = int(col.numel())
    vals = torch.ones(nnz, dtype=torch.int64, device=dev)
    L = gb.Matrix.from_csr(gb.INT64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    dL = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    flops = 2 * int(dL[col.to(torch.int64) & 0xFFFFFFFF].sum())
    nnz_c = None

    def single():
        return L.mxm(L, semiring=gb.INT64.PLUS_PAIR, mask=L).reduce_int()
    if world == 1:
        run = single
        my_nnz = nnz
    else:
        tb = cx.gdist.flop_balanced_row_blocks(rowptr, col, world)
        Lrows, my_nnz = _rows_slice(cx, gb.INT64, rowptr, col, vals, tb[rank], tb[rank + 1], n)

        def run():
            return cx.gdist.triangle_count(cx.comm, Lrows, L)
    tri, first_s = first_run(cx, run)                                            # first run: row binning buffers, pool warm-up
    times = []
    for _ in range(5):
        cx.barrier(); t = time.perf_counter()
        tri = run()
        cx.barrier(); times.append(cx.max_over_ranks(time.perf_counter() - t))
    best = sorted(times)[len(times) // 2]                                        # the median run, like the headline (round 3 reported the minimum)
    plan = gb.last_kernel_plan()
    # A and M streams (this graph's, once), the B-row entries of every product, C written (nnz(C) <= nnz(M))
    alg_bytes = 2 * (nnz * 4 + (n + 1) * 4) + (flops // 2) * 4 + nnz * 12
    traffic, traffic_source = static_traffic("spgemm_pmc_traffic.json") if (world == 1 and scale == 22) else (None, None)
    out = {"workload": f"triangle count R-MAT-{scale}: L.mxm(L, PLUS_PAIR, mask=L).reduce_int() (BASELINE.json configs[3])" + (f", rows of L and of the mask in {world} flop-balanced blocks, L replicated" if world > 1 else ""),
           "nnz_L": nnz, "triangles": int(tri), "flops": flops, "seconds": round(best, 5), "first_run_seconds": round(first_s, 5), "first_run_builds": "row-binning buffers, the result's allocations (pool misses)",
           "GFLOPS": round(flops / best / 1e9, 1), "dtype": "int64",
           "roofline": roof(alg_bytes, best, traffic=traffic, traffic_source=traffic_source,
                            note="B-row entries are counted once per product although part of them is served by L2 / Infinity Cache; whole job (all ranks)"),
           "kernel": plan}
    if rank == 0:
        if world > 1:
            single(); torch.cuda.synchronize(); t = time.perf_counter(); one = single(); torch.cuda.synchronize()
            out["single_gpu_seconds_on_rank0"] = round(time.perf_counter() - t, 5)
            out["parity_vs_single_gpu"] = "bit-exact" if int(one) == int(tri) else f"MISMATCH (single {one})"
        if not cx.args.no_cpu_baseline:
            from oracle import oracle as O
            t = time.perf_counter(); otri = O.fast_tricount(rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32)); cpu_s = time.perf_counter() - t
            out["parity_vs_oracle"] = "bit-exact" if int(tri) == int(otri) else f"MISMATCH (oracle {otri})"
            out["cpu_baseline"] = {"seconds": round(cpu_s, 3), "GFLOPS": round(flops / cpu_s / 1e9, 2), "cores": O.num_threads(), "kind": "port", "sample": "one pass of oracle fast_tricount on the same L"}
    cx.barrier()
    return out


def bench_bfs(cx, scale):
    """configs[2]: the reference's loop (demo/Introduction-to-GraphBLAS-with-Python.ipynb:4301-4313), level vector checked bit for
    bit.  N > 1: entry-balanced row blocks, the frontier gathered as one bit per vertex (pygraphblas_amd.dist.bfs_levels)."""
    gb, rmat, torch, dev, np, world, rank = cx.gb, cx.rmat, cx.torch, cx.dev, cx.np, cx.world, cx.rank
    n = 1 << scale
    rowptr, col = rmat.csr_torch(scale, dev, seed=42, symmetric=True, drop_self_loops=True)
    nnz = int(col.numel())
    vals = torch.ones(nnz, dtype=torch.bool, device=dev)
    A = gb.Matrix.from_csr(gb.BOOL, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    deg = rowptr[1:] - rowptr[:-1]
    src = int(torch.argmax(deg))
    plans = []

    loops = cx.loops

    def single(record=False):
        # (the kernel names are collected in a run of their own: a `last_kernel_plan()` call and a string split per level are the harness's, not the loop's)
        if record:
            del plans[:]
        return loops.bfs(A, src, plans=plans if record else None)
    if world == 1:
        run = single
    else:
        bounds = cx.gdist.balanced_row_blocks((rowptr.to(torch.int64) & 0xFFFFFFFF).cpu().numpy(), world)
        Arows, _ = _rows_slice(cx, gb.BOOL, rowptr, col, vals, bounds[rank], bounds[rank + 1], n)

        def run():
            return cx.gdist.bfs_levels(cx.comm, Arows, n, bounds, src)
    _, first_s = first_run(cx, run)                                              # first run builds the cached transpose / row heads
    times = []
    for _ in range(7):
        cx.barrier(); t = time.perf_counter(); v, depth = run(); cx.barrier(); times.append(cx.max_over_ranks(time.perf_counter() - t))
    best = sorted(times)[len(times) // 2]                                        # the median run (round 3 reported the minimum)
    lev_mine, _ = v.to_dense_arrays()
    out = {"workload": f"BFS R-MAT-{scale} BOOL LOR_LAND, the reference's vxm loop (BASELINE.json configs[2])" + (f", {world} entry-balanced row blocks, bit frontier" if world > 1 else ""),
           "nnz": nnz, "source": src, "depth": depth, "seconds": round(best, 7), "seconds_runs": [round(x, 7) for x in times], "first_run_seconds": round(first_s, 5),
           "first_run_builds": "the transpose the pull levels read (cached with the matrix), the row heads of the masked pull, the loop's vectors (pool misses)", "dtype": "bool"}
    # parity and the rate need the whole level vector: every rank holds the graph, so the single-GPU loop gives it
    v1, d1 = single(record=True); lev, _ = v1.to_dense_arrays()
    if world > 1:
        ok = bool(d1 == depth and np.array_equal(lev[bounds[rank]:bounds[rank + 1]], lev_mine))
        flag = torch.tensor([int(ok)], dtype=torch.int64); cx.tdist.all_reduce(flag, op=cx.tdist.ReduceOp.MIN)
        out["parity_vs_single_gpu"] = "bit-exact level vector (every rank's slice)" if int(flag[0]) else "MISMATCH"
    else:
        out["kernels_per_level"] = list(plans)
    reached = lev > 0
    degh = deg.cpu().numpy().astype(np.int64)
    edges = int(degh[reached].sum()); vr = int(reached.sum())
    # E_r·4 + V_r·8 + n·1 (level write) + n/8·2·D (bitmap read + write per level)
    alg_bytes = edges * 4 + vr * 8 + n + (n // 8) * 2 * depth
    out.update({"reached": vr, "edges_reachable": edges, "GTEPS": round(edges / best / 1e9, 2),
                "roofline": roof(alg_bytes, best, note="E_r*4 + V_r*8 + n + n/8*2*D (SURVEY.md 8d): the whole loop, host round trips of the loop condition included")})
    if rank == 0 and not cx.args.no_cpu_baseline:
        from oracle import oracle as O
        t = time.perf_counter(); olev, odepth = O.fast_bfs(rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32), src); cpu_s = time.perf_counter() - t
        out["parity_vs_oracle"] = "bit-exact level vector" if bool(np.array_equal(olev, lev) and odepth == d1) else "MISMATCH"
        out["cpu_baseline"] = {"seconds": round(cpu_s, 4), "GTEPS": round(edges / cpu_s / 1e9, 3), "cores": O.num_threads(), "kind": "port", "sample": "one pass of oracle fast_bfs"}
    cx.barrier()
    return out


def bench_pagerank(cx, scale, bounds, iters, name):
    """configs[4]: gap/prmark.py's loop in FP32.  One rank: the reference's loop as written (pygraphblas_amd.loops.pagerank: the
    adjacency matrix, desc T0).  N > 1: the row-partitioned form (pygraphblas_amd.dist.pagerank).  A fixed number of iterations is
    timed, then the loop runs to convergence as the reference does."""
    gb, rmat, gdist, torch, dev, world, rank = cx.gb, cx.rmat, cx.gdist, cx.torch, cx.dev, cx.world, cx.rank
    n = 1 << scale
    r0, r1 = bounds[rank], bounds[rank + 1]
    if world == 1:
        rowptr, col = rmat.csr_torch(scale, dev, seed=42)
        nnz = int(col.numel())
        ones = torch.ones(nnz, dtype=torch.float32, device=dev)
        A = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (ones.data_ptr(), nnz), device=True)
        deg = (rowptr[1:] - rowptr[:-1]).to(torch.float32)
        pres = (deg > 0).to(torch.uint8)
        del rowptr, col, ones
        torch.cuda.empty_cache()

        def degrees():
            return gb.Vector.from_dense_array((deg.data_ptr(), n), gb.FP32, present=pres.data_ptr(), device=True)

        def run(fixed):
            return cx.loops.pagerank(A, degrees(), fixed_iterations=fixed)
    else:
        rowptr, col = rmat.csr_torch(scale, dev, seed=42, transpose=True, row_range=(r0, r1))        # rows of A'
        nnz = int(col.numel())
        ones = torch.ones(nnz, dtype=torch.float32, device=dev)
        (rpd, cd, vd), (rpo, co, vo) = gdist.split_csr_columns(rowptr, col, r0, r1, ones)
        Dm = gb.Matrix.from_csr(gb.FP32, r1 - r0, n, rpd.data_ptr(), cd.data_ptr(), (vd.data_ptr(), int(cd.numel())), device=True)
        Om = gb.Matrix.from_csr(gb.FP32, r1 - r0, n, rpo.data_ptr(), co.data_ptr(), (vo.data_ptr(), int(co.numel())), device=True)
        del rowptr, col, ones, rpd, cd, vd, rpo, co, vo
        rpa, ca = rmat.csr_torch(scale, dev, seed=42, row_range=(r0, r1))                             # out-degrees of the owned vertices
        deg = (rpa[1:] - rpa[:-1]).to(torch.float32)
        pres = (deg > 0).to(torch.uint8)
        del rpa, ca
        torch.cuda.empty_cache()

        def degrees():
            return gb.Vector.from_dense_array((deg.data_ptr(), r1 - r0), gb.FP32, present=pres.data_ptr(), device=True)

        def run(fixed):
            return gdist.pagerank(cx.comm, Dm, Om, degrees(), n, bounds, fixed_iterations=fixed)
    (r3, _, _), first_s = first_run(cx, lambda: run(3))                                               # plans, pool warm-up; and the state after 3 iterations for the parity leg
    r3 = r3.to_dense_arrays()[0] if world == 1 else None
    times = []
    for _ in range(5):
        cx.barrier(); t = time.perf_counter()
        r, its, rdiff = run(iters)
        cx.barrier(); times.append(cx.max_over_ranks(time.perf_counter() - t))
    sec = sorted(times)[2]                                                                            # the median of five timed runs (every run's figure is in the line)
    plan = gb.last_kernel_plan()
    nnz_total = int(cx.sum_over_ranks(float(nnz)))
    conv = run(None)                                                                                  # to convergence, as the reference runs it
    # algorithmic bytes of one iteration on this rank, fully fused: the pattern product nnz·4 + (nrows+1)·4 + ncols·4 (w) + nrows·4 (r written;
    # `r[:] = teleport` and the accumulate fold into that store) ; w = t / d reads t, d and writes w (3 × 4 B per owned vertex);
    # t = |t − r| reads t, r and writes t (3 × 4 B), its sum costs no traffic
    nb = r1 - r0
    alg = nnz * 4 + (nb + 1) * 4 + n * 4 + nb * 4 + 6 * nb * 4
    side = {}
    if world == 1 and rank == 0 and not cx.args.no_cpu_baseline:
        # parity + CPU baseline (rank 0, N = 1): gap/prmark.py's iteration on the host cores with the oracle's PLUS_SECOND products over
        # the rows of A' — three iterations in doubles against the GPU's state after three, then a bounded number in FP32, timed
        from oracle import oracle as O
        np = cx.np
        trp, tcol = rmat.csr_torch(scale, dev, seed=42, transpose=True)
        rp, ci = trp.cpu().numpy().view(np.uint32), tcol.cpu().numpy().view(np.uint32)
        del trp, tcol
        torch.cuda.empty_cache()
        degh = deg.cpu().numpy().astype(np.float64)
        dd = np.where(degh > 0, degh / 0.85, 1.0); rr = np.full(n, 1.0 / n); tt = np.zeros(n)
        for _ in range(3):
            tt, rr = rr, tt
            yy, _ = O.fast_spmv(rp, ci, None, np.where(degh > 0, tt / dd, 0.0), semiring="PLUS_SECOND")
            rr = (1 - 0.85) / n + yy
        dev_rel = float(np.max(np.abs(r3.astype(np.float64) - rr) / rr))
        side["parity_vs_oracle"] = ("ok" if dev_rel <= 1e-6 else "MISMATCH") + f" (rank vector after 3 iterations vs the oracle's loop in doubles: max relative deviation {dev_rel:.2e}, tolerance 1e-6)"
        dd32 = dd.astype(np.float32); t32 = rr.astype(np.float32); has = degh > 0
        reps, t0 = 0, time.perf_counter()
        while reps < 20 and (reps < 2 or time.perf_counter() - t0 < 8.0):
            w32 = np.where(has, t32 / dd32, np.float32(0)).astype(np.float32)
            y32, _ = O.fast_spmv(rp, ci, None, w32, semiring="PLUS_SECOND")
            r32 = (np.float32((1 - 0.85) / n) + y32).astype(np.float32); _ = float(np.abs(t32 - r32).sum()); t32 = r32; reps += 1
        cpu_it = (time.perf_counter() - t0) / reps
        side["cpu_baseline"] = {"value": round(2.0 * nnz / cpu_it / 1e9, 3), "unit": "GFLOP/s", "ms_per_iteration": round(cpu_it * 1e3, 2), "cores": O.num_threads(), "kind": "port",
                                "sample": f"{reps} iterations of the same loop on the host: oracle fast_spmv_plus_second_fp32 (OpenMP, {O.num_threads()} threads) + numpy vector steps"}
        del rp, ci
    return {**side, "workload": f"PageRank R-MAT-{scale} FP32, gap/prmark.py loop (PLUS_SECOND, accum PLUS, w = t/d, |t-r| reduced) on {world} row block(s) (BASELINE.json configs[4])",
            "dtype": "f32", "iterations_timed": its, "ms_per_iteration": round(sec / its * 1e3, 4), "first_run_seconds": round(first_s, 5),
            "first_run_builds": "three iterations on a fresh matrix: the transpose (desc T0 pulls along A'), kernel X's panel plan, the chains' code objects (hipRTC, first process only)", "ms_per_iteration_runs": [round(x / its * 1e3, 4) for x in times], "GFLOPS": round(2.0 * nnz_total * its / sec / 1e9, 1), "nnz": nnz_total,
            "iterations_to_converge": conv[1], "rdiff": float(conv[2]), "kernel": plan,
            "roofline": roof(alg, sec / its, note="per iteration on this rank: product nnz*4+(nrows+1)*4+ncols*4+nrows*4, plus 6 vector streams of 4 B per owned vertex (w = t/d; t = |t-r|) — what a fully fused iteration must move")}


def bench_sssp(cx, scale):
    """MIN_PLUS at scale: the reference's shortest-path loop (demo/Intro-Prez.ipynb:1034-1045) on the directed R-MAT-`scale`
    with INT64 weights in [1, 255], distances checked bit for bit against the oracle's loop (sweep count included)."""
    gb, rmat, torch, dev, np = cx.gb, cx.rmat, cx.torch, cx.dev, cx.np
    n = 1 << scale
    rowptr, col = rmat.csr_torch(scale, dev, seed=42, drop_self_loops=True)
    nnz = int(col.numel())
    vals = (rmat.values_torch(nnz, dev, seed=47) * 255.0).to(torch.int64) + 1
    A = gb.Matrix.from_csr(gb.INT64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    deg = (rowptr[1:] - rowptr[:-1])
    src = int(torch.argmax(deg))
    plans = []
    _, first_s = first_run(cx, lambda: cx.loops.sssp(A, src))                     # cached transpose, plans
    times = []
    for _ in range(5):
        del plans[:]
        torch.cuda.synchronize(); t = time.perf_counter(); v, sweeps = cx.loops.sssp(A, src, plans=plans); torch.cuda.synchronize(); times.append(time.perf_counter() - t)
    best = sorted(times)[len(times) // 2]                                         # the median run (round 3 reported the minimum)
    last_plan = cx.gb.last_kernel_plan()                                          # (of the last call of the loop's last run: the one-pass iseq launches no product, so this is the last sweep's)
    gd, gp = v.to_dense_arrays()
    # algorithmic bytes: per sweep the edges leaving the operand's entries (weight + column: 12 B) + the operand's entries (8 B) +
    # the output read and written with its presence bytes (2·9 B per vertex) + the loop's dup and iseq (4 streams of 9 B)
    degh = deg.cpu().numpy().astype(np.int64)
    acc = [0]

    def account(vs):
        _, ps = vs.to_dense_arrays()
        acc[0] += int(degh[ps != 0].sum()) * 12 + int((ps != 0).sum()) * 8 + n * 9 * 6
    cx.loops.sssp(A, src, before_sweep=account)                                   # (untimed pass)
    alg = acc[0]
    out = {"workload": f"SSSP R-MAT-{scale} INT64 MIN_PLUS: v<accum MIN> = v MIN_PLUS A until nothing changes (demo/Intro-Prez.ipynb:1034-1045)", "nnz": nnz, "source": src,
           "sweeps": sweeps, "reached": int((gp != 0).sum()), "seconds": round(best, 5), "first_run_seconds": round(first_s, 5),
           "first_run_builds": "the transpose (vxm pulls along A'), the value range of A, kernel W's then kernel X's plan of the transpose, the chains' code objects",
           "ms_per_sweep": round(best / sweeps * 1e3, 3), "dtype": "int64", "kernels_per_sweep": list(plans), "last_sweep_plan": last_plan.strip(),
           "stored_value_bytes_note": "the algorithmic bytes count a weight as 8 bytes (INT64); kernel X's plan keeps weights that all fit 16 bits as an int16 plane (plan string: values=int16), so 2 of them move",
           "roofline": roof(alg, best, note="sum over sweeps of E_s*12 + V_s*8 (edges leaving / entries of the operand) + 6 vector streams of 9 B per vertex (output read + write, dup, iseq)")}
    if not cx.args.no_cpu_baseline:
        from oracle import oracle as O
        t = time.perf_counter()
        dist, pres, osweeps = O.fast_sssp(rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32), vals.cpu().numpy(), src)
        cpu_s = time.perf_counter() - t
        ok = bool(osweeps == sweeps and np.array_equal(gp != 0, pres != 0) and np.array_equal(gd[pres != 0], dist[pres != 0]))
        out["parity_vs_oracle"] = "bit-exact distances, same sweep count" if ok else f"MISMATCH (oracle sweeps {osweeps})"
        out["cpu_baseline"] = {"seconds": round(cpu_s, 3), "cores": O.num_threads(), "kind": "port", "sample": "one pass of oracle fast_sssp (includes its transpose)"}
    return out


def bench_aa(cx, scale=18, edgefactor=16):
    """The north star's general SpGEMM: the UNMASKED A @ A (lib.GrB_mxm with mask = NULL, pygraphblas/matrix.py:2572-2583) on the symmetric
    R-MAT-`scale`, FP64 PLUS_TIMES, through the two-pass (symbolic + numeric) LDS-hash Gustavson of grb_spgemm_hash.hpp.  Median of five runs;
    sampled rows of the result (every numeric bin, the hub rows) against the oracle's Gustavson rows; the oracle's rows timed as the CPU baseline."""
    gb, rmat, torch, dev, np, lib = cx.gb, cx.rmat, cx.torch, cx.dev, cx.np, cx.lib
    n = 1 << scale
    rowptr, col = rmat.csr_torch(scale, dev, seed=42, edgefactor=edgefactor, symmetric=True, drop_self_loops=True)
    nnz = int(col.numel())
    vals = rmat.values_torch(nnz, dev, seed=45).to(torch.float64) + 0.5
    A = gb.Matrix.from_csr(gb.FP64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    dA = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    products = int(dA[col.to(torch.int64) & 0xFFFFFFFF].sum())
    Cm, first_s = first_run(cx, lambda: A.mxm(A, semiring=gb.FP64.PLUS_TIMES))   # first run: the result's 35 GB come from hipMalloc, later ones from the pool
    times = []
    for _ in range(5):
        Cm = None
        torch.cuda.synchronize(); t = time.perf_counter()
        Cm = A.mxm(A, semiring=gb.FP64.PLUS_TIMES)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t)
    sec = sorted(times)[2]
    plan = gb.last_kernel_plan()
    nc = Cm.nvals
    # SURVEY.md 8d, unmasked form: A once (12 B per entry + row pointers), one B-row entry (column + FP64 value) per product, C written
    alg = nnz * 12 + (n + 1) * 4 + products * 12 + nc * 12 + (n + 1) * 4
    out = {"workload": f"A @ A (unmasked GrB_mxm, two-pass LDS-hash Gustavson) R-MAT-{scale} (edge factor {edgefactor}) symmetric FP64 PLUS_TIMES", "n": n, "nnz_A": nnz, "products": products, "nnz_C": nc,
           "seconds": round(sec, 5), "seconds_runs": [round(x, 5) for x in times], "first_run_seconds": round(first_s, 5),
           "first_run_builds": "the result's arrays and the symbolic pass's temporaries straight from hipMalloc (later runs: the pool)",
           "GFLOPS": round(2.0 * products / sec / 1e9, 1), "dtype": "f64", "kernel": plan,
           "roofline": roof(alg, sec, note="nnz(A)*12 + products*12 (B-row entries, column + value, counted once per product) + nnz(C)*12 + 2*(n+1)*4")}
    if not cx.args.no_cpu_baseline:
        from oracle import oracle as O
        crp = torch.empty(n + 1, dtype=torch.int32, device=dev); ccol = torch.empty(nc, dtype=torch.int32, device=dev); cval = torch.empty(nc, dtype=torch.float64, device=dev)
        gb.base.check(lib.GrBX_Matrix_export_CSR(Cm._h, C.c_void_p(crp.data_ptr()), C.c_void_p(ccol.data_ptr()), C.c_void_p(cval.data_ptr()), C.c_int(1)))
        Cm = None
        rp64 = crp.to(torch.int64) & 0xFFFFFFFF
        lens = (rp64[1:] - rp64[:-1]).cpu().numpy()
        order = np.argsort(lens, kind="stable"); order = order[lens[order] > 0]
        rows = np.unique(np.concatenate([order[np.linspace(0, len(order) - 1, 400).astype(np.int64)], order[-32:], np.random.default_rng(9).choice(n, 200, replace=False)])).astype(np.uint32)
        r = torch.as_tensor(rows.astype(np.int64), device=dev)
        b = rp64[r]; ln = rp64[r + 1] - b
        off = torch.zeros(len(rows) + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(ln, 0)
        idx = torch.arange(int(off[-1]), device=dev, dtype=torch.int64) - torch.repeat_interleave(off[:-1], ln) + torch.repeat_interleave(b, ln)
        gc, gv = ccol[idx].cpu().numpy().view(np.uint32), cval[idx].cpu().numpy()
        del crp, ccol, cval, idx
        rp_h, col_h, val_h = rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32), vals.cpu().numpy()
        woff, wc, wv, wprod = O.fast_mxm_rows(rp_h, col_h, val_h, rows)
        ok = bool(np.array_equal(off.cpu().numpy(), woff) and np.array_equal(gc, wc) and np.allclose(gv, wv, rtol=1e-6, atol=0.0))
        out["parity_vs_oracle"] = (f"ok ({len(rows)} sampled rows of C — spread over the rows' lengths, the 32 longest, 200 random; {len(wc)} entries: pattern exact, values rtol 1e-6)"
                                   if ok else "MISMATCH on the sampled rows")
        # CPU baseline: the oracle's Gustavson rows (dense accumulator per thread) on random rows until ~8 s are spent
        rng = np.random.default_rng(10); done_p, reps, t0 = 0, 0, time.perf_counter()
        while reps < 40 and (reps < 1 or time.perf_counter() - t0 < 8.0):
            rs = rng.choice(n, 4096, replace=False).astype(np.uint32)
            done_p += int(O.fast_mxm_rows(rp_h, col_h, val_h, rs)[3].sum()) * 2          # (count pass + fill pass: every product is formed twice)
            reps += 1
        cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(2.0 * done_p / cpu_s / 1e9, 3), "unit": "GFLOP/s", "cores": O.num_threads(), "kind": "port",
                               "sample": f"{reps * 4096} random rows of the same product ({done_p} products) with oracle fast_mxm_rows_plus_times_fp64 (OpenMP, {O.num_threads()} threads, dense accumulator per thread)",
                               "seconds_extrapolated_to_the_whole_product": round(products / (done_p / cpu_s), 2)}
    Cm = None
    return out


def bench_masked_fp64_deterministic(cx, scale):
    """configs[3]'s shape on floating-point values: C<L> = L (+.x) L, FP64 PLUS_TIMES, in the default mode (LDS / HBM atomics as they land) and with
    GRB_MI355X_DETERMINISTIC=1 (exact 128-bit integer accumulators: the same bits in every run, the exactly rounded sums)."""
    import ctypes as C
    gb, rmat, torch, dev = cx.gb, cx.rmat, cx.torch, cx.dev
    n = 1 << scale
    rowptr, col = rmat.csr_torch(scale, dev, seed=42, symmetric=True, drop_self_loops=True, lower=True)
    nnz = int(col.numel())
    vals = rmat.values_torch(nnz, dev, seed=46) + 0.5
    L = gb.Matrix.from_csr(gb.FP64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    dL = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    products = int(dL[col.to(torch.int64) & 0xFFFFFFFF].sum())

    def run():
        Cm = L.mxm(L, semiring=gb.FP64.PLUS_TIMES, mask=L, desc=cx.gb.descriptor.S)
        Cm.nvals
        return Cm

    def values_of(Cm):
        v = torch.empty(Cm.nvals, dtype=torch.float64, device=dev)
        gb.base.check(gb.lib.GrBX_Matrix_export_CSR(Cm._h, None, None, C.c_void_p(v.data_ptr()), C.c_int(1)))
        return v
    res = {}
    saved = os.environ.get("GRB_MI355X_DETERMINISTIC")
    try:
        for mode in ("default", "deterministic"):
            if mode == "deterministic": os.environ["GRB_MI355X_DETERMINISTIC"] = "1"
            else: os.environ.pop("GRB_MI355X_DETERMINISTIC", None)
            run(); times = []
            for _ in range(5):
                torch.cuda.synchronize(); t = time.perf_counter(); Cm = run(); torch.cuda.synchronize(); times.append(time.perf_counter() - t)
            v = values_of(Cm); again = values_of(run())
            res[mode] = {"seconds": round(sorted(times)[2], 5), "plan": gb.last_kernel_plan().strip(), "same_bits_twice": bool(torch.equal(v.view(torch.int64), again.view(torch.int64)))}
            if mode == "default": v0 = v
            else: res[mode]["agrees_with_default_rtol_1e-10"] = bool(torch.allclose(v, v0, rtol=1e-10, atol=0.0))
            del again
    finally:
        if saved is None: os.environ.pop("GRB_MI355X_DETERMINISTIC", None)
        else: os.environ["GRB_MI355X_DETERMINISTIC"] = saved
    alg = 2 * (nnz * 12 + (n + 1) * 4) + products * 4 + int(v0.numel()) * 12
    return {"workload": f"masked product on FP64 values R-MAT-{scale}: L.mxm(L, PLUS_TIMES, mask=L), default mode against GRB_MI355X_DETERMINISTIC=1 (median of 5 each)",
            "nnz_L": nnz, "products": products, "entries": int(v0.numel()), "dtype": "f64", "default": res["default"], "deterministic": res["deterministic"],
            "deterministic_over_default": round(res["deterministic"]["seconds"] / res["default"]["seconds"], 3),
            "roofline": roof(alg, res["deterministic"]["seconds"], note="the deterministic mode's time; the triangle count's convention (SURVEY.md 8d): nnz(L)*(4+8)*2 + row pointers + products*4 (the column of every B-row entry; a value is read only by a hit) + entries*12")}


def bench_bc(cx, scale):
    """The batched betweenness centrality of gap/bcmark.py:16-67 (tools/bc_algorithm.py restates the driver statement for statement), ns = 4
    sources of maximum degree on the directed R-MAT-`scale`, FP32: masked PLUS_FIRST GrB_mxm per level forwards and backwards.  The step the
    north star's SpGEMM serves is the frontier product; the whole algorithm is timed, the forward sweep's products beside it."""
    gb, rmat, torch, dev, np = cx.gb, cx.rmat, cx.torch, cx.dev, cx.np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bc_algorithm import bc as bc_full
    from pygraphblas_amd import descriptor as D
    n = 1 << scale; ns = 4
    rowptr, col = rmat.csr_torch(scale, dev, seed=42, drop_self_loops=True)
    nnz = int(col.numel()); vals = torch.ones(nnz, dtype=torch.float32, device=dev)
    A = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    AT = A.transpose()
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    sources = [int(x) for x in torch.argsort(deg, descending=True, stable=True)[:ns].cpu()]
    _, first_s = first_run(cx, lambda: bc_full(gb, sources, AT, A))               # cached transposes, pool
    times = []; sizes = []
    for _ in range(5):
        del sizes[:]
        torch.cuda.synchronize(); t = time.perf_counter(); cent, depth = bc_full(gb, sources, AT, A, sizes=sizes); torch.cuda.synchronize(); times.append(time.perf_counter() - t)
    sec = sorted(times)[2]
    plan = gb.last_kernel_plan()
    cv = cent.to_dense_arrays()[0]
    # the forward sweep's products alone (frontier<!paths,replace> = frontier (+).first A per level), synchronised per level
    paths = gb.Matrix.dense(gb.FP32, ns, n, 0); frontier = gb.Matrix.sparse(gb.FP32, ns, n)
    for i, s in enumerate(sources):
        paths[i, s] = 1; frontier[i, s] = 1
    step_s = []
    for _ in range(depth + 1):
        torch.cuda.synchronize(); t = time.perf_counter()
        frontier.mxm(A, out=frontier, mask=paths, semiring=gb.FP32.PLUS_FIRST, desc=D.RC)
        torch.cuda.synchronize(); step_s.append(time.perf_counter() - t)
        if frontier.nvals == 0:
            break
        paths.assign_matrix(frontier, accum=gb.FP32.PLUS)
    # algorithmic bytes, a lower bound for ANY implementation of the driver: every source's forward and backward sweep each read the edges
    # leaving / entering its reached vertices once (4 B column per edge: FIRST ignores A's values), the reached vertices' path counts and
    # dependencies (2 x 8 B each per sweep), and the centrality is written once
    pv, pp = None, None
    reached = np.zeros(ns, np.int64); edges = np.zeros(ns, np.int64)
    degh = deg.cpu().numpy()
    prp, pci, pvals = paths.to_csr()
    for s in range(ns):
        seg = slice(int(prp[s]), int(prp[s + 1])); hit = pci[seg][pvals[seg] != 0].astype(np.int64)
        reached[s] = len(hit); edges[s] = int(degh[hit].sum())
    alg = int(2 * edges.sum() * 4 + 2 * reached.sum() * 16 + n * 4)
    out = {"workload": f"batched betweenness centrality, gap/bcmark.py:16-67, R-MAT-{scale} directed, ns = {ns} sources of maximum degree, FP32 (masked PLUS_FIRST GrB_mxm per level)",
           "nnz": nnz, "depth": depth, "frontier_nvals": list(sizes), "seconds": round(sec, 5), "seconds_runs": [round(x, 5) for x in times], "first_run_seconds": round(first_s, 5),
           "first_run_builds": "the transposes the pull levels read, row heads, the dense ns x n batches (pool misses)", "dtype": "f32",
           "parity_tolerance_note": "centrality rtol 1e-4: FP32 sums of the driver against an FP64 restatement (north star's 1e-6 holds for FP64 / FP32 products against same-precision references; here the reference is wider)",
           "forward_products_ms": [round(x * 1e3, 3) for x in step_s], "kernel": plan,
           "roofline": roof(alg, sec, note="lower bound of any implementation: per source, forward and backward sweep each read the edges of the reached vertices once (4 B) and "
                                           "their path counts / dependencies (2 x 8 B); centrality written once.  The driver's own formulation moves ns x n dense batches per level on top")}
    if not cx.args.no_cpu_baseline:
        from oracle import oracle as O
        rpt, colt = AT.to_csr()[:2]
        t = time.perf_counter()
        want, odepth, olv = O.fast_bc(rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32), rpt, colt, sources)
        cpu_s = time.perf_counter() - t
        ok = bool(depth == odepth and list(sizes) == list(olv) and np.allclose(cv.astype(np.float64), want, rtol=1e-4, atol=1e-3))
        out["parity_vs_oracle"] = "ok (depth and every level's frontier size exact, centrality rtol 1e-4 vs the FP64 restatement)" if ok else f"MISMATCH (oracle depth {odepth}, levels {olv})"
        out["cpu_baseline"] = {"seconds": round(cpu_s, 3), "cores": O.num_threads(), "kind": "port", "sample": "one pass of oracle fast_bc_batch (the same algorithm on dense ns x n batches of doubles, OpenMP)"}
    return out


if __name__ == "__main__":
    main()
