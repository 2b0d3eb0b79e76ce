#!/bin/bash
out=gpurun_out/r4ac; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for L in base m8; do
  if [ $L = m8 ]; then export GRB_MI355X_LIB=$GRAFT_REPO_ROOT/pygraphblas_amd/libgrb_m8.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$L -o kt -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels a,4o > $out/kt_$L.log 2>&1
  grep -h '^{' $out/kt_$L.log | cut -c1-200
done
python - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for L in ("base", "m8"):
    for f in glob.glob(f"{out}/kt_{L}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_xp_merge_wide" in r["Name"] or "k_spmv_tiles" in r["Name"]: print(L, r["Name"].split("(")[0][-60:], r["Calls"], round(float(r["AverageNs"])/1e3, 1))
PY
find $out -name "*kernel_trace.csv" -delete
