#!/bin/bash
out=gpurun_out/r4z; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_subpanels_gpu.py tests/test_mxv_vxm_gpu.py tests/test_baseline_configs_gpu.py -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $out/tests.log | grep -v "Librccl\|Hostname"
GRB_MI355X_XC_VERIFY=1 timeout 600 python tools/r4_subpanel_probe.py --skip-b --subpanels 1 --oracle > $out/probeA.log 2>&1
timeout 600 python tools/r4_subpanel_probe.py --skip-a --pr-subpanels a,8o > $out/probeB.log 2>&1
grep -h '^{' $out/probeA.log $out/probeB.log | cut -c1-260; grep -h "plane check" $out/probeA.log | head -2
for S in 1; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python tools/r4_subpanel_probe.py --skip-b --subpanels 1 > $out/kt.log 2>&1
done
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "grb::" in n and any(k in n for k in ("k_spmv_tiles", "k_xp_merge", "k_xp_hot")):
            print(f'   {n.split("(")[0][-72:]:72s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us')
PY
find $out -name "*kernel_trace.csv" -delete
