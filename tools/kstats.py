#!/usr/bin/env python3
"""Print the grb:: kernels of a rocprofv3 --kernel-trace --stats output directory (calls, average and total time)."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
        print(f'{r["Name"].split("(")[0][-110:]:110s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:10.1f} us total {float(r["TotalDurationNs"]) / 1e6:9.2f} ms')
