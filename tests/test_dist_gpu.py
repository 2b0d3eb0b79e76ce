"""The row-partitioned workloads (pygraphblas_amd/dist.py) with the HIP kernels running in every rank.

Two ranks share the one GPU of the box: RCCL refuses two ranks on one device, so the slices travel through the "host"
transport (torch.distributed / gloo) — everything else (partitioning, diagonal / off-diagonal split, the products, the bit
frontier, the reductions) is the code an 8-GPU run executes.  Each workload is compared with the single-process result:
bit-exact for the BFS level vector and the INT64 triangle count, 1e-6 relative for the FP32 PageRank vector.
The library's own RCCL path is exercised with a communicator of one rank (init, all-reduce, allgatherv, bits).
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALE = 14


def _single_process_reference(gb):
    """PageRank vector, BFS levels and the triangle count of R-MAT-14 computed by one process (the loops of the reference)."""
    from pygraphblas_amd import rmat, dist as gdist, descriptor as D
    n = 1 << SCALE
    # PageRank on A' (rows of the transpose), exactly the distributed code with a world of one
    comm = gdist.Comm(0, 1)
    rp, col = rmat.csr_numpy(SCALE, transpose=True)
    At = gb.Matrix.from_csr(gb.FP32, n, n, rp, col, np.ones(len(col), np.float32))
    empty = gb.Matrix.sparse(gb.FP32, n, n)
    rpa, _ = rmat.csr_numpy(SCALE)
    deg = np.diff(rpa.astype(np.int64))
    d = gb.Vector.from_arrays(np.flatnonzero(deg).astype(np.uint64), deg[deg > 0].astype(np.float32), n, gb.FP32)
    r, its, rdiff = gdist.pagerank(comm, At, empty, d, n, [0, n])
    pr = r.to_dense_arrays()[0]
    # BFS
    rps, cols = rmat.csr_numpy(SCALE, symmetric=True, drop_self_loops=True)
    A = gb.Matrix.from_csr(gb.BOOL, n, n, rps, cols, np.ones(len(cols), np.bool_))
    src = int(np.argmax(np.diff(rps.astype(np.int64))))
    v = gb.Vector.sparse(gb.UINT8, n); q = gb.Vector.sparse(gb.BOOL, n); q[src] = True
    level = 1
    while q.reduce_bool() and level <= n:
        v.assign_scalar(level, mask=q)
        v.vxm(A, mask=v, out=q, desc=D.RC)
        level += 1
    lev = v.to_dense_arrays()[0]
    # triangles
    rpl, coll = rmat.csr_numpy(SCALE, symmetric=True, drop_self_loops=True, lower=True)
    L = gb.Matrix.from_csr(gb.INT64, n, n, rpl, coll, np.ones(len(coll), np.int64))
    tri = L.mxm(L, semiring=gb.INT64.PLUS_PAIR, mask=L).reduce_int()
    return {"pr": pr, "its": its, "lev": lev, "depth": level - 1, "src": src, "tri": tri}


FAKE_RCCL = os.path.join(ROOT, "tests", "libfake_rccl.so")


def _worker(rank, world, port, q, transport="host"):
    try:
        sys.path.insert(0, ROOT)
        os.environ["GRB_MI355X_DEVICE"] = "0"
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # the host driver supports dmabuf IPC only
        if transport == "rccl":
            os.environ["GRB_MI355X_RCCL"] = FAKE_RCCL                      # grb_dist.cpp binds the stand-in instead of librccl
        import torch
        import torch.distributed as tdist
        tdist.init_process_group("gloo", rank=rank, world_size=world)
        import pygraphblas_amd as gb
        from pygraphblas_amd import rmat, dist as gdist
        assert gb.device_info()["ok"]
        n = 1 << SCALE

        def share(ident):
            box = [ident]
            tdist.broadcast_object_list(box, src=0)
            return box[0]
        comm = gdist.Comm(rank, world, transport=transport, share=share, tdist=tdist)
        out = {"transport": gdist.bound_transport()}
        # ---- PageRank: rows of A' balanced by entries, split into diagonal / off-diagonal columns
        rp_all, _ = rmat.csr_numpy(SCALE, transpose=True)
        bounds = gdist.balanced_row_blocks(rp_all.astype(np.int64), world)
        r0, r1 = bounds[rank], bounds[rank + 1]
        rp, col = rmat.csr_numpy(SCALE, transpose=True, row_range=(r0, r1))
        (rpd, cd, _), (rpo, co, _) = gdist.split_csr_columns(torch.from_numpy(rp.view(np.int32)), torch.from_numpy(col.view(np.int32)), r0, r1)
        mk = lambda p, c: gb.Matrix.from_csr(gb.FP32, r1 - r0, n, p.numpy().view(np.uint32), c.numpy().view(np.uint32), np.ones(c.numel(), np.float32))
        Dm, Om = mk(rpd, cd), mk(rpo, co)
        assert Dm.nvals + Om.nvals == len(col)
        rpa, _ = rmat.csr_numpy(SCALE, row_range=(r0, r1))
        deg = np.diff(rpa.astype(np.int64))
        d = gb.Vector.from_arrays(np.flatnonzero(deg).astype(np.uint64), deg[deg > 0].astype(np.float32), r1 - r0, gb.FP32)
        r, its, rdiff = gdist.pagerank(comm, Dm, Om, d, n, bounds)
        out["pr"] = (r0, r1, r.to_dense_arrays()[0], its)
        # ---- BFS: rows of the symmetric graph, frontier gathered as bits
        rps_all, _ = rmat.csr_numpy(SCALE, symmetric=True, drop_self_loops=True)
        b2 = gdist.balanced_row_blocks(rps_all.astype(np.int64), world)
        s0, s1 = b2[rank], b2[rank + 1]
        rps, cols = rmat.csr_numpy(SCALE, symmetric=True, drop_self_loops=True, row_range=(s0, s1))
        Arows = gb.Matrix.from_csr(gb.BOOL, s1 - s0, n, rps, cols, np.ones(len(cols), np.bool_))
        src = int(np.argmax(np.diff(rps_all.astype(np.int64))))
        v_loc, depth = gdist.bfs_levels(comm, Arows, n, b2, src)
        out["bfs"] = (s0, s1, v_loc.to_dense_arrays()[0], depth)
        # ---- triangles: L replicated, rows balanced by the flop bound
        rpl, coll = rmat.csr_numpy(SCALE, symmetric=True, drop_self_loops=True, lower=True)
        b3 = gdist.flop_balanced_row_blocks(torch.from_numpy(rpl.view(np.int32)), torch.from_numpy(coll.view(np.int32)), world)
        t0, t1 = b3[rank], b3[rank + 1]
        L = gb.Matrix.from_csr(gb.INT64, n, n, rpl, coll, np.ones(len(coll), np.int64))
        e0, e1 = int(rpl[t0]), int(rpl[t1])
        Lrows = gb.Matrix.from_csr(gb.INT64, t1 - t0, n, (rpl[t0:t1 + 1] - rpl[t0]).astype(np.uint32), coll[e0:e1], np.ones(e1 - e0, np.int64))
        out["tri"] = (gdist.triangle_count(comm, Lrows, L), t0, t1)
        comm.close()
        q.put((rank, out))
        tdist.destroy_process_group()
    except Exception as e:          # noqa: BLE001 — surface the failure in the parent
        import traceback
        q.put((rank, {"error": traceback.format_exc() + repr(e)}))


def _ensure_fake_rccl():
    src = os.path.join(ROOT, "tests", "fake_rccl.cpp")
    if not os.path.exists(FAKE_RCCL) or os.path.getmtime(FAKE_RCCL) < os.path.getmtime(src):
        import subprocess
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", "-o", FAKE_RCCL, src, "-lrt"])


def test_two_ranks_on_one_gpu_match_the_single_process_results(gb, gpu):
    _two_ranks(gb, "host")


def test_two_ranks_on_one_gpu_through_the_library_exchange_path(gb, gpu):
    """The same three workloads with the slices moving through the LIBRARY's exchange (grb_dist.cpp: grouped ncclSend / ncclRecv on its
    second stream, ready / done events, presence bytes, the bit frontier, ncclAllReduce) between two real ranks.  RCCL refuses two ranks on
    one device, so the nine entry points it binds come from tests/libfake_rccl.so, which moves the DEVICE pointers it is handed between
    the two processes with hipIpc memory handles — everything but xGMI."""
    _ensure_fake_rccl()
    res = _two_ranks(gb, "rccl")
    for r in (0, 1):
        assert res[r]["transport"].endswith("libfake_rccl.so"), res[r]["transport"]


def test_four_ranks_on_one_gpu_through_the_library_exchange_path(gb, gpu):
    """World size 4 (what the driver's N = 4 run executes per rank, minus xGMI): every rank sends its slice to three peers and receives three in
    one group — the rotating peer order of GrBX_Vector_allgatherv_start, four slices of the bit frontier, a four-way all-reduce."""
    _ensure_fake_rccl()
    res = _two_ranks(gb, "rccl", world=4)
    for r in range(4):
        assert res[r]["transport"].endswith("libfake_rccl.so"), res[r]["transport"]


def test_eight_ranks_on_one_gpu_through_the_library_exchange_path(gb, gpu):
    """World size 8 — the first real multi-GPU run will be the driver's N = 8 — at small scale: seven peers per rank in one send / receive group, eight
    slices of the bit frontier, an eight-way all-reduce, eight entry-balanced row blocks of a skewed graph (the first block holds a handful of hub rows)."""
    _ensure_fake_rccl()
    res = _two_ranks(gb, "rccl", world=8)
    for r in range(8):
        assert res[r]["transport"].endswith("libfake_rccl.so"), res[r]["transport"]


def _two_ranks(gb, transport, world=2):
    import torch.multiprocessing as mp
    ref = _single_process_reference(gb)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000) + (11 if transport == "rccl" else 0) + 3 * (world - 2)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(60)
    for r in range(world):
        assert "error" not in res[r], res[r]["error"]
    n = 1 << SCALE
    pr = np.zeros(n, np.float32); lev = np.zeros(n, np.uint8)
    for r in range(world):
        a, b, x, its = res[r]["pr"]; pr[a:b] = x
        assert its == ref["its"]
        a, b, x, depth = res[r]["bfs"]; lev[a:b] = x
        assert depth == ref["depth"]
    for r in range(world - 1):
        assert res[r]["pr"][1] == res[r + 1]["pr"][0]                                  # contiguous blocks ...
    assert 0 < res[0]["pr"][1] < n                                                      # ... more than one of them non-empty
    assert np.allclose(pr, ref["pr"], rtol=1e-6, atol=0.0)
    assert np.array_equal(lev, ref["lev"])                                              # bit-exact level vector
    assert all(res[r]["tri"][0] == ref["tri"] for r in range(world))                    # INT64, exact
    assert 0 < res[0]["tri"][2] < n
    return res


def test_library_rccl_path_with_a_communicator_of_one(gb, gpu):
    """GrBX_dist_init / allreduce / allgatherv / allgatherv_bits through RCCL itself (one rank: what a 1-GPU box allows)."""
    import ctypes as C
    lib = gb.lib
    ident = C.create_string_buffer(128)
    assert lib.GrBX_dist_unique_id(ident, C.c_int(128)) == 0
    assert lib.GrBX_dist_init(C.c_int(0), C.c_int(1), ident, C.c_int(128)) == 0
    try:
        x = np.array([3.5, -1.25], np.float64)
        assert lib.GrBX_dist_allreduce(x.ctypes.data_as(C.c_void_p), C.c_uint64(2), C.c_void_p(gb.FP64._h), C.c_void_p(gb.FP64.PLUS.get_op())) == 0
        assert x.tolist() == [3.5, -1.25]
        k = np.array([2 ** 40 + 7], np.int64)
        assert lib.GrBX_dist_allreduce(k.ctypes.data_as(C.c_void_p), C.c_uint64(1), C.c_void_p(gb.INT64._h), C.c_void_p(gb.INT64.PLUS.get_op())) == 0
        assert int(k[0]) == 2 ** 40 + 7
        n = 1000
        rng = np.random.default_rng(5)
        idx = np.sort(rng.choice(n, 300, replace=False)).astype(np.uint64)
        loc = gb.Vector.from_arrays(idx, rng.random(300), n, gb.FP64)
        full = gb.Vector.dense(gb.FP64, n, fill=0.0)
        b = np.array([0, n], np.uint64)
        assert lib.GrBX_Vector_allgatherv(full._h, loc._h, b.ctypes.data_as(C.c_void_p), C.c_int(1)) == 0
        assert lib.GrBX_Vector_device_touch(full._h) == 0
        fi, fx = full.to_arrays(); li, lx = loc.to_arrays()
        assert np.array_equal(fi, li) and np.array_equal(fx, lx)
        ql = gb.Vector.from_arrays(idx, rng.integers(0, 2, 300).astype(np.bool_), n, gb.BOOL)
        qf = gb.Vector.sparse(gb.BOOL, n)
        assert lib.GrBX_Vector_allgatherv_bits(qf._h, ql._h, b.ctypes.data_as(C.c_void_p)) == 0
        qi, qx = qf.to_arrays(); ei, ex = ql.to_arrays()
        assert np.array_equal(qi, ei[ex]) and qx.all()                                   # only the `true` entries travel
    finally:
        assert lib.GrBX_dist_finalize() == 0


@pytest.mark.parametrize("scaling", ["weak", "strong", "weak-through-the-library-exchange-path"])
def test_bench_py_runs_its_two_rank_path_on_one_gpu(gpu, scaling):
    """`bench.py --gpus 2` — what the driver launches for the scaling curve — end to end on this box: two ranks on GPU 0
    (BENCH_DEVICE_OVERRIDE), the exchange through the host transport (RCCL cannot put two ranks on one device), at R-MAT-17.
    The JSON line must parse, carry the N = 2 fields in both scaling modes, the other mode's SpMV with per-phase times, and the
    distributed mxm / bfs / pagerank sub-objects with their parity fields (triangle count bit-exact against the single-GPU
    count and the oracle, level vector bit-exact on every rank's slice, PageRank converging in the single-process count)."""
    import json, subprocess
    env = dict(os.environ, BENCH_DEVICE_OVERRIDE="0", BENCH_TRANSPORT="host", MASTER_ADDR="127.0.0.1")
    fake = scaling.endswith("exchange-path")           # the third case: `Comm("rccl")` bound to tests/libfake_rccl.so (hipIpc between the two ranks)
    if fake:
        _ensure_fake_rccl()
        env.update(BENCH_TRANSPORT="rccl", GRB_MI355X_RCCL=FAKE_RCCL, HSA_ENABLE_IPC_MODE_LEGACY="0")
        scaling = "weak"
    port = 29800 + (os.getpid() % 1000) + (7 if scaling == "strong" else 0) + (13 if fake else 0)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--blocks", "2", "--scale", "17", "--pr-scale", "16", "--scaling", scaling],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == scaling and d["value"] > 0
    assert d["config"]["n"] == (1 << 18 if scaling == "weak" else 1 << 17)
    assert len(d["ms_per_step_blocks"]) == 2 and ("rccl-abi (fake, hipIpc)" if fake else "host copies") in d["config"]["transport"]
    other = d["spmv_strong" if scaling == "weak" else "spmv_weak"]
    assert other["n"] == (1 << 17 if scaling == "weak" else 1 << 18) and other["GFLOPS"] > 0
    for ph in (d["phases"], other["phases"]):
        assert ph["exchange_ms"] > 0 and ph["diag_ms"] > 0 and ph["offdiag_ms"] > 0
    assert d["roofline"]["bound"] == "hbm"
    assert d["mxm"]["parity_vs_single_gpu"] == "bit-exact" and d["mxm"]["parity_vs_oracle"] == "bit-exact" and d["mxm"]["roofline"]["frac"] > 0
    assert d["bfs"]["parity_vs_single_gpu"].startswith("bit-exact") and d["bfs"]["parity_vs_oracle"].startswith("bit-exact") and d["bfs"]["roofline"]["frac"] > 0
    assert d["pagerank"]["iterations_to_converge"] > 5 and d["pagerank"]["dtype"] == "f32" and d["pagerank"]["roofline"]["frac"] > 0
    assert d["pagerank_scale25"]["iterations_to_converge"] > 5 and "R-MAT-16" in d["pagerank_scale25"]["workload"]
