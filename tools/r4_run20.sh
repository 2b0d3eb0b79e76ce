#!/bin/bash
out=gpurun_out/r4u; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for W in 0; do
  GRB_MI355X_XM_WIDE=$W timeout 300 python tools/r4_subpanel_probe.py --skip-b --subpanels 1 --oracle > $out/probe_w$W.log 2>&1
  grep -h '^{' $out/probe_w$W.log | cut -c1-330
  GRB_MI355X_XM_WIDE=$W timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_w$W -o kt -- python tools/r4_subpanel_probe.py --skip-b --subpanels 1 > $out/kt_w$W.log 2>&1
done
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for W in ("0", "1"):
    print("==== XM_WIDE =", W)
    for f in glob.glob(f"{out}/kt_w{W}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Name"]
            if "grb::" in n and any(k in n for k in ("k_spmv_tiles", "k_xp_merge", "k_xp_hot", "k_xp_lpt", "k_xp_sweep")):
                print(f'   {n.split("(")[0][-70:]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us')
PY
find $out -name "*kernel_trace.csv" -size +3M -delete
