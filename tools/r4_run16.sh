#!/bin/bash
out=gpurun_out/r4r; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py tests/test_mxv_vxm_gpu.py -x -q -k "sssp or min_plus or holes or nan" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $out/tests.log
timeout 300 python tools/sssp_probe.py > $out/sssp_probe.log 2>&1; cat $out/sssp_probe.log | grep -v amdgpu
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python tools/sssp_probe.py > $out/kt.log 2>&1
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:22]:
        print(f'   {r["Name"].split("(")[0][-80:]:80s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
find $out -name "*kernel_trace.csv" -size +3M -delete
