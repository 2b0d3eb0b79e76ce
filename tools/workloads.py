#!/usr/bin/env python3
"""BASELINE.json configs[2] and [3] at scale: push/pull BFS (repeated GrB_vxm) and triangle counting
(masked GrB_mxm) on R-MAT, checked bit-exactly against the oracle's typed CPU loops.  Experiment /
measurement harness around the product; prints one JSON line per workload."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat, descriptor as D

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22)
ap.add_argument("--what", default="bfs,tc,pr,bc")
ap.add_argument("--aa-edgefactor", type=int, default=16)
ap.add_argument("--aa-scale", type=int, default=18)
ap.add_argument("--aa-methods", default="hash,esc")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--no-check", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)
n = 1 << args.scale
lib = gb.lib


def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, time.perf_counter() - t


def bfs(A, start):
    """The reference's loop, verbatim in structure (demo/Introduction-to-GraphBLAS-with-Python.ipynb cell 31)."""
    v = gb.Vector.sparse(gb.UINT8, A.nrows)
    q = gb.Vector.sparse(gb.BOOL, A.nrows)
    q[start] = True
    level = 1
    plans = []
    while q.reduce_bool() and level <= A.nrows:
        v.assign_scalar(level, mask=q)
        v.vxm(A, mask=v, out=q, desc=D.RC)
        plans.append(gb.last_kernel_plan())
        level += 1
    return v, level - 1, plans


if "bfs" in args.what:
    rowptr, col = rmat.csr_torch(args.scale, dev, seed=42, symmetric=True, drop_self_loops=True)
    nnz = col.numel()
    vals = torch.ones(nnz, dtype=torch.bool, device=dev)
    A = gb.Matrix.from_csr(gb.BOOL, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    deg = (rowptr[1:] - rowptr[:-1])
    src = int(torch.argmax(deg))
    (v, depth, plans), t0 = timed(lambda: bfs(A, src))     # first run builds cached transposes / plans
    best = 1e9
    for _ in range(args.reps):
        (v, depth, plans), t = timed(lambda: bfs(A, src)); best = min(best, t)
    lev, _ = v.to_dense_arrays()
    reached = int((lev > 0).sum())
    edges_reached = int(deg.cpu().numpy()[lev > 0].sum())
    out = {"workload": f"BFS R-MAT-{args.scale} BOOL LOR_LAND vxm loop", "n": n, "nnz": nnz, "source": src, "depth": depth, "reached": reached,
           "seconds": round(best, 5), "first_run_seconds": round(t0, 4), "GTEPS": round(edges_reached / best / 1e9, 3), "plans": plans}
    if not args.no_check:
        from oracle import oracle as O
        rp, ci = rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32)
        t = time.perf_counter(); olev, odepth = O.fast_bfs(rp, ci, src); out["cpu_seconds"] = round(time.perf_counter() - t, 4)
        out["parity"] = "bit-exact" if (np.array_equal(olev, lev) and odepth == depth) else "MISMATCH"
        out["cpu_threads"] = O.num_threads()
    print(json.dumps(out), flush=True)
    del A, rowptr, col, vals

if "tc" in args.what:
    rowptr, col = rmat.csr_torch(args.scale, dev, seed=42, symmetric=True, drop_self_loops=True, lower=True)
    nnz = col.numel()
    vals = torch.ones(nnz, dtype=torch.int64, device=dev)
    L = gb.Matrix.from_csr(gb.INT64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    dL = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    flops = 2 * int(dL[col.to(torch.int64) & 0xFFFFFFFF].sum())
    def tc():
        return L.mxm(L, semiring=gb.INT64.PLUS_PAIR, mask=L).reduce_int()
    tri, t0 = timed(tc)
    best = 1e9
    for _ in range(args.reps):
        tri, t = timed(tc); best = min(best, t)
    alg_bytes = 2 * (nnz * 4 + (n + 1) * 4) + (flops // 2) * 4
    out = {"workload": f"triangle count R-MAT-{args.scale}: L.mxm(L, PLUS_PAIR, mask=L).reduce_int()", "n": n, "nnz_L": nnz, "triangles": tri,
           "flops": flops, "seconds": round(best, 5), "first_run_seconds": round(t0, 4), "GFLOPS": round(flops / best / 1e9, 2),
           "GBps_algorithmic": round(alg_bytes / best / 1e9, 1), "plan": gb.last_kernel_plan()}
    if not args.no_check:
        from oracle import oracle as O
        rp, ci = rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32)
        t = time.perf_counter(); otri = O.fast_tricount(rp, ci); out["cpu_seconds"] = round(time.perf_counter() - t, 3)
        out["parity"] = "bit-exact" if otri == tri else f"MISMATCH (oracle {otri})"
        out["cpu_threads"] = O.num_threads()
    print(json.dumps(out), flush=True)


if "tcfp" in args.what:
    # the masked product on floating-point values: C<L> = L (+.x) L, FP64 — default mode (atomics as they land) against the deterministic mode
    # (GRB_MI355X_DETERMINISTIC=1: exact 128-bit integer accumulators, grb_exact.hpp), same bits from run to run
    rowptr, col = rmat.csr_torch(args.scale, dev, seed=42, symmetric=True, drop_self_loops=True, lower=True)
    nnz = col.numel()
    vals = rmat.values_torch(nnz, dev, seed=46) + 0.5
    L = gb.Matrix.from_csr(gb.FP64, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    dL = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    products = int(dL[col.to(torch.int64) & 0xFFFFFFFF].sum())
    def values_of(Cm):
        nv = Cm.nvals
        cval = torch.empty(nv, dtype=torch.float64, device=dev)
        gb.base.check(lib.GrBX_Matrix_export_CSR(Cm._h, None, None, C.c_void_p(cval.data_ptr()), C.c_int(1)))
        return cval
    def run():
        Cm = L.mxm(L, semiring=gb.FP64.PLUS_TIMES, mask=L, desc=D.S); Cm.nvals; return Cm
    res = {}
    for mode in ("default", "deterministic"):
        if mode == "deterministic": os.environ["GRB_MI355X_DETERMINISTIC"] = "1"
        Cm, _ = timed(run); best = 1e9
        for _ in range(args.reps):
            Cm, t = timed(run); best = min(best, t)
        v = values_of(Cm)
        again = values_of(run())
        res[mode] = {"seconds": round(best, 5), "plan": gb.last_kernel_plan().strip(), "same_bits_twice": bool(torch.equal(v.view(torch.int64), again.view(torch.int64)))}
        if mode == "default": v0 = v
        else: res[mode]["agrees_with_default_rtol_1e-10"] = bool(torch.allclose(v, v0, rtol=1e-10, atol=0.0)); res[mode]["entries"] = int(v.numel())
    os.environ.pop("GRB_MI355X_DETERMINISTIC", None)
    alg_bytes = 2 * (nnz * 12 + (n + 1) * 4) + products * 12
    print(json.dumps({"workload": f"masked product on FP64 values R-MAT-{args.scale}: L.mxm(L, PLUS_TIMES, mask=L)", "n": n, "nnz_L": nnz, "products": products,
                      "default": res["default"], "deterministic": res["deterministic"],
                      "deterministic_over_default": round(res["deterministic"]["seconds"] / res["default"]["seconds"], 3),
                      "GBps_algorithmic_deterministic": round(alg_bytes / res["deterministic"]["seconds"] / 1e9, 1)}), flush=True)


def pagerank(A, d, damping, itermax):
    """gap/prmark.py:8-30 with the modern descriptor name (descriptor.T0 for the stale `TransposeA`)."""
    from pygraphblas_amd import Vector, FP32
    n = A.nrows
    r = Vector.sparse(FP32, n)
    t = Vector.sparse(FP32, n)
    d.assign_scalar(damping, accum=FP32.DIV)
    r[:] = 1.0 / n
    teleport = (1 - damping) / n
    tol = 1e-4
    rdiff = 1.0
    its = 0
    for i in range(itermax):
        temp = t; t = r; r = temp
        w = t / d
        r[:] = teleport
        A.mxv(w, out=r, accum=FP32.PLUS, semiring=FP32.PLUS_SECOND, desc=D.T0)
        t -= r
        t.apply(FP32.ABS, out=t)
        rdiff = t.reduce_float()
        its = i + 1
        if rdiff <= tol:
            break
    return r, its, rdiff


if "pr" in args.what:
    rowptr, col = rmat.csr_torch(args.scale, dev, seed=42)
    nnz = col.numel()
    vals = torch.ones(nnz, dtype=torch.float32, device=dev)
    A = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    def run():
        d = A.reduce_vector()                      # out-degree (PLUS_MONOID over rows); dangling rows have no entry
        return pagerank(A, d, 0.85, 100)
    (r, its, rdiff), t0 = timed(run)
    best = 1e9
    for _ in range(args.reps):
        (r, its, rdiff), t = timed(run); best = min(best, t)
    rv, rp_ = r.to_dense_arrays()
    out = {"workload": f"PageRank R-MAT-{args.scale} FP32 (gap/prmark.py loop: PLUS_SECOND mxv with T0 + accum, 5 vector ops/iter)", "n": n, "nnz": nnz,
           "iterations": its, "rdiff": float(rdiff), "seconds": round(best, 5), "first_run_seconds": round(t0, 4), "ms_per_iteration": round(best / its * 1e3, 4),
           "plan": gb.last_kernel_plan()}
    if not args.no_check:
        import scipy.sparse as sp
        rp, ci = rowptr.cpu().numpy().view(np.uint32).astype(np.int64), col.cpu().numpy().view(np.uint32).astype(np.int64)
        At = sp.csr_matrix((np.ones(nnz, np.float64), ci, rp), shape=(n, n)).T.tocsr()
        deg = np.diff(rp).astype(np.float64)
        t = time.perf_counter()
        dd = np.where(deg > 0, deg / 0.85, np.nan); rr = np.full(n, 1.0 / n); tt = np.zeros(n); k = 0
        for i in range(100):
            tt, rr = rr, tt
            w = np.where(deg > 0, tt / dd, 0.0)
            rr = (1 - 0.85) / n + At @ w
            k = i + 1
            if np.abs(tt - rr).sum() <= 1e-4:
                break
        out["cpu_seconds_scipy_fp64"] = round(time.perf_counter() - t, 3)
        rel = np.abs(rv.astype(np.float64) - rr) / np.abs(rr)
        out["parity"] = {"iterations_equal": bool(k == its), "max_rel_err_vs_fp64": float(rel.max()), "all_present": bool(rp_.all())}
    print(json.dumps(out), flush=True)


if "bc" in args.what:
    # the batched-frontier step of gap/bcmark.py:16-44: frontier<!paths,replace> = frontier (+).first A, ns = 4 sources, FP32
    ns = 4
    rowptr, col = rmat.csr_torch(args.scale, dev, seed=42, symmetric=True, drop_self_loops=True)
    nnz = col.numel()
    vals = torch.ones(nnz, dtype=torch.float32, device=dev)
    A = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    sources = torch.argsort(deg, descending=True, stable=True)[:ns]
    prp = (torch.arange(ns + 1, device=dev, dtype=torch.int64) * n).to(torch.int32)
    pcol = torch.arange(n, device=dev, dtype=torch.int32).repeat(ns)
    def fresh():
        pv = torch.zeros(ns * n, dtype=torch.float32, device=dev); pv[torch.arange(ns, device=dev) * n + sources] = 1.0
        paths = gb.Matrix.from_csr(gb.FP32, ns, n, prp.data_ptr(), pcol.data_ptr(), (pv.data_ptr(), ns * n), device=True)
        frp = torch.arange(ns + 1, device=dev, dtype=torch.int32); fv = torch.ones(ns, dtype=torch.float32, device=dev)
        frontier = gb.Matrix.from_csr(gb.FP32, ns, n, frp.data_ptr(), sources.to(torch.int32).data_ptr(), (fv.data_ptr(), ns), device=True)
        return paths, frontier
    levels = []
    for rep in range(2):                                        # second pass: warm pool / cached structures
        paths, frontier = fresh(); levels = []
        for depth in range(n):
            torch.cuda.synchronize(); t = time.perf_counter()
            frontier.mxm(A, out=frontier, mask=paths, semiring=gb.FP32.PLUS_FIRST, desc=D.RC)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            nv = frontier.nvals
            levels.append({"depth": depth, "seconds": round(dt, 5), "frontier_nvals": nv, "plan": gb.last_kernel_plan()})
            if nv == 0:
                break
            paths = paths.eadd(frontier, gb.FP32.PLUS)
    print(json.dumps({"workload": f"BC batched-frontier step R-MAT-{args.scale}, ns={ns}: frontier<!paths,replace> = frontier PLUS_FIRST A (gap/bcmark.py:16-44)", "n": n, "nnz": nnz,
                      "seconds_all_levels": round(sum(l["seconds"] for l in levels), 5), "levels": levels}), flush=True)


if "aa" in args.what:
    # the unmasked product A @ A (lib.GrB_mxm without a mask, pygraphblas/matrix.py:2572-2583): two-pass LDS-hash Gustavson vs expand/sort/compress
    S = args.aa_scale; m = 1 << S
    rowptr, col = rmat.csr_torch(S, dev, seed=42, edgefactor=args.aa_edgefactor, symmetric=True, drop_self_loops=True)
    nnz = col.numel(); vals = torch.ones(nnz, dtype=torch.float64, device=dev)
    A = gb.Matrix.from_csr(gb.FP64, m, m, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    dA = (rowptr[1:] - rowptr[:-1]).to(torch.int64); products = int(dA[col.to(torch.int64) & 0xFFFFFFFF].sum())
    res = {}
    aa_methods = tuple(args.aa_methods.split(","))
    for method in aa_methods:
        os.environ["GRB_MI355X_SPGEMM"] = method
        best = 1e9; Cm = None
        for _ in range(3):
            Cm = None                                                     # (the previous result goes back to the pool: the next call reuses its 35 GB instead of a fresh hipMalloc)
            torch.cuda.synchronize(); base = C.c_size_t(0); lib.GrBX_memory_in_use(C.byref(base)); t = time.perf_counter()
            Cm = A.mxm(A, semiring=gb.FP64.PLUS_TIMES); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        res[method] = {"seconds": round(best, 4), "GFLOPS": round(2 * products / best / 1e9, 1), "nnz_C": Cm.nvals, "plan": gb.last_kernel_plan()}
        del Cm
    os.environ.pop("GRB_MI355X_SPGEMM")
    res["temporaries"] = {"hash_note": "16 B per entry of the rows that go through the LDS tables only (rows beyond them are written in place by the LDS dense path)", "esc_bytes_per_product": 40, "esc_bytes": 40 * min(products, 1 << 27)}
    # algorithmic bytes (unmasked form): A once, one B-row entry (column + FP64 value) per product, C written
    nc = res["hash"]["nnz_C"]
    alg = nnz * 12 + (m + 1) * 4 + products * 12 + nc * 12 + (m + 1) * 4
    for method in aa_methods:
        a = alg / res[method]["seconds"] / 1e9
        res[method]["roofline"] = {"bound": "hbm", "achieved": round(a, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(a / 8000.0, 4), "algorithmic_bytes": alg,
                                   "note": "nnz(A)*12 + products*12 (B-row entries, column + value) + nnz(C)*12 + 2*(n+1)*4; the hash path also writes and sorts 16 B per entry of C"}
    print(json.dumps({"workload": f"A @ A (unmasked GrB_mxm) R-MAT-{S} symmetric FP64 PLUS_TIMES", "n": m, "nnz_A": nnz, "products": products, **res}), flush=True)


if "bcfull" in args.what:
    # the whole batched BC of gap/bcmark.py:16-67 (tools/bc_algorithm.py), ns = 4 sources of maximum degree, directed R-MAT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bc_algorithm import bc as bc_full
    rowptr, col = rmat.csr_torch(args.scale, dev, seed=42, drop_self_loops=True)
    nnz = col.numel(); vals = torch.ones(nnz, dtype=torch.float32, device=dev)
    A = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
    AT = A.transpose()
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    sources = [int(x) for x in torch.argsort(deg, descending=True, stable=True)[:4].cpu()]
    best = 1e9
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter(); cent, depth = bc_full(gb, sources, AT, A); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    cv = cent.to_dense_arrays()[0]
    # parity: the oracle's restatement of the same algorithm in doubles (oracle/grb_oracle.c fast_bc_batch) — depth exact, values to 1e-4 (the driver is FP32)
    from oracle import oracle as O
    rpt, colt = AT.to_csr()[:2]
    want, odepth, _ = O.fast_bc(rowptr.cpu().numpy().view(np.uint32), col.cpu().numpy().view(np.uint32), rpt, colt, sources)
    ok = depth == odepth and bool(np.allclose(cv.astype(np.float64), want, rtol=1e-4, atol=1e-3))
    print(json.dumps({"workload": f"batched betweenness centrality, gap/bcmark.py:16-67, R-MAT-{args.scale} directed, ns=4", "n": n, "nnz": nnz, "depth": depth, "seconds": round(best, 4),
                      "max_centrality": float(cv.max()), "parity_vs_oracle": "ok (depth exact, centrality rtol 1e-4 vs the FP64 restatement)" if ok else "MISMATCH", "plan": gb.last_kernel_plan()}), flush=True)
