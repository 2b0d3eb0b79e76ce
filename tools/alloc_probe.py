#!/usr/bin/env python3
"""Measurement harness: what hipMalloc / hipFree cost on this box as a function of the size (the cold part of a plan build)."""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipFree(None)
p = C.c_void_p()
for mb in (1, 16, 64, 256, 512, 1024):
    ts = []
    for rep in range(3):
        t = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), C.c_size_t(mb << 20)); t1 = time.perf_counter()
        hip.hipMemset(p, 0, C.c_size_t(mb << 20)); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
        hip.hipFree(p); t3 = time.perf_counter()
        ts.append((t1 - t, t2 - t1, t3 - t2))
    print(f"{mb:5d} MiB: hipMalloc {min(x[0] for x in ts)*1e3:7.3f} ms  first memset {min(x[1] for x in ts)*1e3:7.3f} ms  hipFree {min(x[2] for x in ts)*1e3:7.3f} ms")
