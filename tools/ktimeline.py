#!/usr/bin/env python3
"""Print the LAST `count` kernels of a rocprofv3 --kernel-trace output directory in launch order: gap to the previous kernel's end, duration, name.
usage: python tools/ktimeline.py <dir> [count]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-cnt:]
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f'gap {gap:8.1f} us  dur {(e - s) / 1e3:8.1f} us  {r["Kernel_Name"].split("(")[0][-90:]}')
    prev = e
