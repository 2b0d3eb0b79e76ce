#!/bin/bash
out=gpurun_out/r4aa; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python tools/workloads.py --what aa --aa-methods hash > $out/aa.log 2>&1
grep '^{' $out/aa.log | cut -c1-400
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:25]:
        print(f'   {r["Name"].split("(")[0][-90:]:90s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
find $out -name "*kernel_trace.csv" -delete
