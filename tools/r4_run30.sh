#!/bin/bash
out=gpurun_out/r4am; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_mxm_gpu.py -x -q -k "hash_spgemm" 2>&1 | tail -4
for L in mi355x; do
GRB_MI355X_LIB=$GRAFT_REPO_ROOT/pygraphblas_amd/libgrb_$L.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/aa_$L -o aa -- python tools/workloads.py --what aa --aa-methods hash > $out/aa_$L.log 2>&1
grep -h '^{' $out/aa_$L.log | cut -c1-330
done
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in sorted(glob.glob(f"{out}/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("spa_", "k_spgemm", "k_hash", "k_spa")) and float(r["AverageNs"]) > 1e6: print(f.split("/")[-2], r["Name"].split("(")[0][-70:], r["Calls"], round(float(r["AverageNs"])/1e3, 1))
PY
find $out -name "*kernel_trace.csv" -delete
