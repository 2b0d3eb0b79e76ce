#!/bin/bash
# round-4 GPU session 4: a table per sub-panel (GRB_MI355X_XOWN), new at-scale parity tests, plan policy
out=gpurun_out/r4d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
GRB_MI355X_XC_VERIFY=1 timeout 600 python tools/r4_subpanel_probe.py --oracle --skip-b --subpanels 1,2o,4o > $out/probeA.log 2>&1; echo "probeA rc=$?"
timeout 900 python tools/r4_subpanel_probe.py --skip-a --pr-subpanels 1,4o,8o > $out/probeB.log 2>&1; echo "probeB rc=$?"
for S in 4o 8o; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$S -o kt -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels $S > $out/kt_$S.log 2>&1
done
timeout 1200 python -m pytest tests/test_baseline_configs_gpu.py tests/test_nonblocking_gpu.py tests/test_mxv_vxm_gpu.py -x -q > $out/tests.log 2>&1; echo "tests rc=$?"
grep -h '^{' $out/probeA.log $out/probeB.log | cut -c1-400
grep -h "column plane check" $out/probeA.log | head -4
tail -n 4 $out/tests.log
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for S in ("4o", "8o"):
    print("==== kernel trace, S =", S)
    for f in glob.glob(f"{out}/kt_{S}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Name"]
            if "grb::" in n and any(k in n for k in ("k_spmv_tiles", "k_xp_merge", "k_xp_hot", "k_vec_chain", "k_xp_sweep")):
                print(f'   {n.split("(")[0][-70:]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us')
PY
find $out -name "*kernel_trace.csv" -size +3M -delete
