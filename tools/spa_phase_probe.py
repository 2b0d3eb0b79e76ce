#!/usr/bin/env python3
"""Measurement harness (not part of the product): where a workgroup of k_spgemm_spa_numeric spends its clocks on A @ A (R-MAT-18, FP64).
Needs the -DSPA_PROFILE build:  make BUILD=build_spa LIB=../libgrb_spa.so XTFLAGS=-DSPA_PROFILE  and  GRB_MI355X_LIB=.../libgrb_spa.so."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat
dev = torch.device("cuda", 0); lib = gb.lib
S = int(sys.argv[1]) if len(sys.argv) > 1 else 18; m = 1 << S
rowptr, col = rmat.csr_torch(S, dev, seed=42, symmetric=True, drop_self_loops=True)
nnz = col.numel(); vals = torch.ones(nnz, dtype=torch.float64, device=dev)
A = gb.Matrix.from_csr(gb.FP64, m, m, rowptr.data_ptr(), col.data_ptr(), (vals.data_ptr(), nnz), device=True)
for _ in range(2):
    Cm = None; torch.cuda.synchronize(); t = time.perf_counter(); Cm = A.mxm(A, semiring=gb.FP64.PLUS_TIMES); torch.cuda.synchronize(); sec = time.perf_counter() - t
buf = np.zeros(1024 * 16, dtype=np.uint64)
rc = lib.GrBX_spa_prof_read_double(buf.ctypes.data_as(C.c_void_p))
p = buf.reshape(1024, 16)[:512].astype(np.float64); p = p[p[:, 7] > 0]
names = ["scan+2 barriers", "rounds (thread 0's wave)", "wait for the other waves", "bitmap scan + 2 barriers", "emission", "last barrier"]
tot = p[:, 7].mean()
phases = p[:, :6].sum(axis=1)
print(json.dumps({"seconds": round(sec, 4), "workgroups": len(p), "kernel_clocks_mean": tot, "kernel_clocks_min": p[:, 7].min(), "kernel_clocks_max": p[:, 7].max(),
                  "direct_phase_clocks_mean (the rest of a workgroup's time is its ranked rows and row set-up)": phases.mean(), "steps_with_products": p[:, 6].mean(), "steps_without": p[:, 8].mean(),
                  "share": {n: round(p[:, k].mean() / tot, 4) for k, n in enumerate(names)},
                  "clocks_per_step": {n: round(p[:, k].sum() / p[:, 6].sum(), 1) for k, n in enumerate(names)}, "plan": gb.last_kernel_plan()}))
