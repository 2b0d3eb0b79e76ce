#!/bin/bash
# round 6: HBM traffic and time of kernel X on INT64 MIN_PLUS (weights 1 ... 255, R-MAT-22) with the int16 value plane and without (GRB_MI355X_XT_NARROW=0)
out=${1:-gpurun_out/r6narrow}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
{
for nar in 1 0; do
  echo "=== GRB_MI355X_XT_NARROW=$nar"
  GRB_MI355X_XT_NARROW=$nar timeout 200 python tools/spmv_probe.py --reps 20 --variants INT64.MIN_PLUS,INT64.PLUS_TIMES --methods auto 2>/dev/null | grep -v "^scale"
  GRB_MI355X_XT_NARROW=$nar PMC_PASSES=2 bash tools/pmc_spmv.sh $out/pmc$nar --variants INT64.MIN_PLUS --methods auto 2>/dev/null | grep -A 5 "k_spmv_tiles<long"
done
} > $out/summary.txt 2>&1
cat $out/summary.txt
find $out -name "*counter_collection.csv" -size +1M -delete
