#!/usr/bin/env python3
"""Measurement harness: the element-wise chain kernel in its variants on n = 2^22 FP32 (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pygraphblas_amd as gb
n = 1 << 22
F = gb.FP32
r = gb.Vector.from_dense_array(np.random.default_rng(0).random(n).astype(np.float32), F)
t = gb.Vector.from_dense_array(np.random.default_rng(1).random(n).astype(np.float32), F)
for _ in range(20):          # A: eadd + abs, flushed by wait (no reduction)
    r.eadd(t, F.MINUS, out=t); t.apply(F.ABS, out=t); t.wait()
for _ in range(20):          # B: eadd + abs + reduce_float (FP32 values, FP64 monoid)
    r.eadd(t, F.MINUS, out=t); t.apply(F.ABS, out=t); t.reduce_float()
for _ in range(20):          # C: eadd alone + reduce_float
    r.eadd(t, F.MINUS, out=t); t.reduce_float()
for _ in range(20):          # D: reduce with the FP32 monoid (no widening)
    r.eadd(t, F.MINUS, out=t); t._reduce_scalar("FP32", __import__("ctypes").c_float, F, F.PLUS_MONOID, None, None)
print("ok")
