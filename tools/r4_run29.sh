#!/bin/bash
out=gpurun_out/r4ad; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_mxm_gpu.py -x -q -k "hash_spgemm" 2>&1 | tail -4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/aa -o aa -- python tools/workloads.py --what aa --aa-methods hash > $out/aa.log 2>&1
grep -h '^{' $out/aa.log | cut -c1-600
export GRB_MI355X_LIB=$GRAFT_REPO_ROOT/pygraphblas_amd/libgrb_m2.so
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_m2 -o kt -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels a > $out/kt_m2.log 2>&1
grep -h '^{' $out/kt_m2.log | cut -c1-200
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(f"{out}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_xp_merge_wide", "k_spmv_tiles", "spa_", "k_spgemm", "k_hash", "k_spa")): print(f.split("/")[-2], r["Name"].split("(")[0][-70:], r["Calls"], round(float(r["AverageNs"])/1e3, 1))
PY
find $out -name "*kernel_trace.csv" -delete
