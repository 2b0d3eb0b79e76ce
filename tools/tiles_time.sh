#!/bin/bash
# measurement harness: per-kernel times of kernel X (tiles, merge, table gather) for the FP64 PLUS_TIMES and FP32 PLUS_SECOND products
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/tiles_time; rm -rf $out; mkdir -p $out
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o x -- python tools/spmv_probe.py --reps 30 --variants FP64.PLUS_TIMES,FP32.PLUS_SECOND --methods auto > $out/probe.txt 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python - $f <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if any(t in n for t in ("k_spmv_tiles","k_xp_merge","k_xp_hot_gather")):
        print(n.replace("void ","").split("(")[0][:74], r["Calls"], "avg %.1f us min %.1f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
grep -i "ms\|GFLOP" $out/probe.txt | head -6
find $out/prof -name "*kernel_trace.csv" -delete
