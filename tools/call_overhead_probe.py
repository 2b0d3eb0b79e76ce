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

# ---- the PageRank loop's calls (gap/prmark.py:17-29) on vectors of 4096 positions: what the host spends between the read-back of one iteration and the first launch of the next
from pygraphblas_amd import loops
FP32 = gb.FP32
P = gb.Matrix.from_arrays(key >> np.uint64(32), key & np.uint64(0xFFFFFFFF), np.ones(len(key), np.float32), n, n, FP32)
d = P.reduce_vector(); d.assign_scalar(0.85, accum=FP32.DIV)
r = gb.Vector.dense(FP32, n, fill=1.0 / n); t = gb.Vector.dense(FP32, n, fill=1.0 / n)
def it_div():
    global w
    w = t / d
def it_fill(): r[:] = 0.15 / n
def it_mxv(): P.mxv(w, out=r, accum=FP32.PLUS, semiring=FP32.PLUS_SECOND, desc=D.T0)
def it_sub():
    global t
    t -= r
def it_abs(): t.apply(FP32.ABS, out=t)
def it_red(): return t.reduce_float()
def it_swap():
    global r, t
    r, t = t, r
def one():
    it_swap(); it_div(); it_fill(); it_mxv(); it_sub(); it_abs(); return it_red()
one(); one(); one()
import time as _t
acc = {k: 0.0 for k in ("div", "fill", "mxv", "sub", "abs", "reduce")}
reps = 2000
for _ in range(reps):
    it_swap()
    for k, f in (("div", it_div), ("fill", it_fill), ("mxv", it_mxv), ("sub", it_sub), ("abs", it_abs), ("reduce", it_red)):
        a = _t.perf_counter(); f(); acc[k] += _t.perf_counter() - a
print("PageRank iteration on 4096 positions, host time per call (us):", {k: round(v / reps * 1e6, 2) for k, v in acc.items()}, "sum", round(sum(acc.values()) / reps * 1e6, 1))

# where the 5-6 us of `r[:] = teleport` go: the mirror's Python, or the library
import ctypes as C
from pygraphblas_amd._capi import u64
ALL = C.cast(gb._capi.handle("GrB_ALL"), C.c_void_p)
fn = lib.GrB_Vector_assign_FP32
cval = C.c_float(0.15 / n); zero = u64(0)
def raw_fill(): fn(r._h, None, None, cval, ALL, zero, None)      # (r: the global, swapped every iteration like the loop's)
for name, f in (("r[:] = c", it_fill), ("r.assign_scalar(c)", lambda: r.assign_scalar(0.15 / n)), ("raw GrB_Vector_assign_FP32", raw_fill)):
    one()
    tot = 0.0
    for _ in range(1000):
        it_swap(); it_div(); a = _t.perf_counter(); f(); tot += _t.perf_counter() - a; it_mxv(); it_sub(); it_abs(); it_red()
    print(f"{name:32s} {tot / 1000 * 1e6:6.2f} us")
fnew = lib.GrB_Vector_new; ffree = lib.GrB_Vector_free
h = C.c_void_p()
a = _t.perf_counter()
for _ in range(2000):
    fnew(C.byref(h), C.c_void_p(FP32._h), u64(n)); ffree(C.byref(h))
print(f"GrB_Vector_new + GrB_Vector_free       {(_t.perf_counter() - a) / 2000 * 1e6:6.2f} us")
