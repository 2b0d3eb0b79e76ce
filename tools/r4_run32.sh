#!/bin/bash
out=gpurun_out/r4ao; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels a > $out/kt.log 2>&1
grep -h '^{' $out/kt.log | cut -c1-200
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]: print(r["Name"].split("(")[0][-80:], r["Calls"], round(float(r["AverageNs"])/1e3, 1))
PY
find $out -name "*kernel_trace.csv" -delete
