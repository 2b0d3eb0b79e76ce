#!/bin/bash
out=gpurun_out/r4ap; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for b in 2 8; do
GRB_MI355X_CHAIN_BPC=$b timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt$b -o kt -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels a > $out/kt$b.log 2>&1
python - $out/kt$b $b <<'PY'
import csv, glob, sys
for f in glob.glob(f"{sys.argv[1]}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_vec_chain" in r["Name"] and int(r["Calls"]) >= 30: print("BPC", sys.argv[2], r["Name"].split("(")[0][-50:], r["Calls"], round(float(r["AverageNs"])/1e3, 1))
PY
done
find $out -name "*kernel_trace.csv" -delete
