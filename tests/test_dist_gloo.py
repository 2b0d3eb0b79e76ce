"""The N>1 path on CPU: two gloo ranks run the same partition + allgatherv code bench.py uses on RCCL."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pygraphblas_amd_rmat import rmat
    import importlib.util
    spec = importlib.util.spec_from_file_location("gdist", os.path.join(ROOT, "pygraphblas_amd", "dist.py"))
    gdist = importlib.util.module_from_spec(spec); spec.loader.exec_module(gdist)
    scale = 10; n = 1 << scale
    bounds = gdist.balanced_row_blocks(gdist.rmat_expected_row_prefix(scale), world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    rp, col = rmat.csr_numpy(scale, row_range=(r0, r1))
    x_all = torch.from_numpy(rmat.values_numpy(n, seed=44))
    full = torch.zeros(n, dtype=torch.float64)
    gdist.allgatherv_into(full, x_all[r0:r1].clone(), bounds, rank, world, dist)
    assert torch.equal(full, x_all)
    full2 = torch.zeros(n, dtype=torch.float64)
    gdist.allgatherv_into(full2, x_all[r0:r1].clone(), bounds, rank, world, dist, mode="broadcast")
    assert torch.equal(full2, x_all)
    # local row block times the gathered vector == the matching rows of the global product
    val = rmat.values_numpy(len(col), seed=43 + rank)
    import scipy.sparse as sp
    y = sp.csr_matrix((val, col.astype(np.int64), rp.astype(np.int64)), shape=(r1 - r0, n)) @ full.numpy()
    nnz = torch.tensor([float(len(col))]); dist.all_reduce(nnz)
    q.put((rank, r0, r1, float(y.sum()), float(nnz[0])))
    dist.destroy_process_group()


def test_two_rank_allgatherv_and_partition():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get() for _ in range(2))
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == 1024
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from pygraphblas_amd_rmat import rmat
    rp, col = rmat.csr_numpy(10)
    assert res[0][4] == len(col)                       # the two blocks hold every entry exactly once
