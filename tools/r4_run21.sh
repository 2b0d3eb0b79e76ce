#!/bin/bash
out=gpurun_out/r4v; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python tools/r4_subpanel_probe.py --skip-b --subpanels 1,2o,4o > $out/probeA.log 2>&1; echo "probeA rc=$?"
timeout 900 python tools/r4_subpanel_probe.py --skip-a --pr-subpanels 1,4o,8o > $out/probeB.log 2>&1; echo "probeB rc=$?"
for S in 4o 8o; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$S -o kt -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels $S > $out/kt_$S.log 2>&1
done
grep -h '^{' $out/probeA.log $out/probeB.log | cut -c1-330
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for S in ("4o", "8o"):
    print("==== kernel trace, S =", S)
    for f in glob.glob(f"{out}/kt_{S}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Name"]
            if "grb::" in n and any(k in n for k in ("k_spmv_tiles", "k_xp_merge", "k_xp_hot", "k_vec_chain")):
                print(f'   {n.split("(")[0][-70:]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us')
PY
find $out -name "*kernel_trace.csv" -size +3M -delete
