#!/bin/bash
out=gpurun_out/r4at; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_subpanels_gpu.py -x -q 2>&1 | tail -2
for L in mi355x b0; do
GRB_MI355X_LIB=$GRAFT_REPO_ROOT/pygraphblas_amd/libgrb_$L.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$L -o kt -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels a > $out/kt_$L.log 2>&1
grep -h '^{' $out/kt_$L.log | tail -1 | cut -c1-150
python - $out/kt_$L $L <<'PY'
import csv, glob, sys
for f in glob.glob(f"{sys.argv[1]}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_xp_merge_wide" in r["Name"] or "grb::k_spmv_tiles" in r["Name"]: print(sys.argv[2], r["Name"].split("(")[0][-60:], r["Calls"], round(float(r["AverageNs"])/1e3, 1))
PY
done
find $out -name "*kernel_trace.csv" -delete
