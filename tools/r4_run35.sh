#!/bin/bash
out=gpurun_out/r4aq; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_subpanels_gpu.py tests/test_mxv_vxm_gpu.py -x -q 2>&1 | tail -2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python tools/r4_subpanel_probe.py --scale 22 --subpanels a --pr-subpanels a > $out/kt.log 2>&1
grep -h '^{' $out/kt.log | cut -c1-260
python - $out <<'PY'
import csv, glob, sys
for f in glob.glob(f"{sys.argv[1]}/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "grb::k_spmv_tiles" in r["Name"] or "k_xp_merge" in r["Name"]: print(r["Name"].split("(")[0][-80:], r["Calls"], round(float(r["AverageNs"])/1e3, 1))
PY
find $out -name "*kernel_trace.csv" -delete
