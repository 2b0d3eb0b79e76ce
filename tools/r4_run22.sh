#!/bin/bash
out=gpurun_out/r4w; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for B in 0 6 8 12 16; do
  GRB_MI355X_CHAIN_BPC=$B timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_b$B -o kt -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels a > $out/kt_b$B.log 2>&1
done
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for B in (0, 6, 8, 12, 16):
    for f in glob.glob(f"{out}/kt_b{B}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Name"]
            if "k_vec_chain" in n: print(f'BPC={B:2d}   {n.split("(")[0][-50:]:50s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us')
PY
find $out -name "*kernel_trace.csv" -delete
