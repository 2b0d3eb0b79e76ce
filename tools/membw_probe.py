#!/usr/bin/env python3
"""Experiment harness (not part of the product): what this MI355X sustains for plain streams (torch kernels)."""
import torch, time
dev = torch.device("cuda", 0)
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / reps * 1e-3
for mb in (256, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.rand(n, device=dev, dtype=torch.float32); y = torch.empty_like(x)
    t = timeit(lambda: torch.sum(x));           print(f"{mb:5d} MB  read  (sum f32)      {n*4/t/1e12:6.2f} TB/s")
    xd = x.view(torch.float64)
    t = timeit(lambda: torch.sum(xd));          print(f"{mb:5d} MB  read  (sum f64)      {n*4/t/1e12:6.2f} TB/s")
    t = timeit(lambda: torch.amax(x));          print(f"{mb:5d} MB  read  (amax f32)     {n*4/t/1e12:6.2f} TB/s")
    t = timeit(lambda: y.copy_(x));             print(f"{mb:5d} MB  copy  (read+write)   {2*n*4/t/1e12:6.2f} TB/s")
    t = timeit(lambda: y.fill_(1.0));           print(f"{mb:5d} MB  write (fill)         {n*4/t/1e12:6.2f} TB/s")
    del x, y, xd
