#!/bin/bash
out=gpurun_out/r4y; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_mxv_vxm_gpu.py tests/test_baseline_configs_gpu.py tests/test_companion_ops_gpu.py tests/test_reference_suite_gpu.py tests/test_dist_gpu.py -x -q -k "not rmat25" > $out/tests.log 2>&1; echo "tests rc=$?"
tail -n 4 $out/tests.log
timeout 300 python tools/bfs_probe.py > $out/bfs_probe.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o kt -- python tools/bfs_probe.py > $out/kt.log 2>&1
cat $out/bfs_probe.log
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(f"{out}/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_spmspv_push<" in r["Kernel_Name"]]
i0 = idx[-1] - 8
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 45]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0)/1e3:9.1f} us  +{(e - s)/1e3:7.1f}  {r["Kernel_Name"].split("(")[0].replace("void ", "")[:90]}')
PY
find $out -name "*kernel_trace.csv" -size +3M -delete
