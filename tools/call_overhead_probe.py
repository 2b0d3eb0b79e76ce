#!/usr/bin/env python3
"""Host time per mirror call of the BFS loop's three calls on TINY operands (the kernels are microseconds: what is left is Python + ctypes + the library's host
code + the launch).  usage: python tools/call_overhead_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pygraphblas_amd as gb
from pygraphblas_amd import descriptor as D
import ctypes as C
n = 4096
rng = np.random.default_rng(1)
I = rng.integers(0, n, 20000); J = rng.integers(0, n, 20000)
key = np.unique(I.astype(np.uint64) << np.uint64(32) | J.astype(np.uint64))
A = gb.Matrix.from_arrays(key >> np.uint64(32), key & np.uint64(0xFFFFFFFF), np.ones(len(key), bool), n, n, gb.BOOL)
v = gb.Vector.sparse(gb.UINT8, n); q = gb.Vector.sparse(gb.BOOL, n); q[0] = True
v.assign_scalar(1, mask=q); v.vxm(A, mask=v, out=q, desc=D.RC); q.reduce_bool()
def timeit(name, fn, reps=3000):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:48s} {(t1 - t) / reps * 1e6:7.2f} us per call (host), {(t2 - t) / reps * 1e6:7.2f} us with the final synchronise")
timeit("assign_scalar(level, mask=q)", lambda: v.assign_scalar(3, mask=q))
timeit("vxm(A, mask=v, out=q, desc=RC)", lambda: v.vxm(A, mask=v, out=q, desc=D.RC))
timeit("reduce_bool()", lambda: q.reduce_bool(), 1000)
timeit("nvals (known)", lambda: A.nvals)
lib = gb.lib
h = v._h
nv = C.c_uint64()
timeit("raw ctypes GrB_Vector_size", lambda: lib.GrB_Vector_size(C.byref(nv), h))
timeit("python no-op lambda", lambda: None)
