#!/bin/bash
# differential fuzzing with the round's library on the GPU box (tools/fuzz_parity.py, tools/fuzz_companions.py)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6fuzz; mkdir -p $out
{
echo "differential fuzzing on the MI355X with the round-6 library (tools/fuzz_parity.py: mxv / vxm / mxm against the oracle; tools/fuzz_companions.py: the companion operations against the Python model)"
timeout 200 python tools/fuzz_parity.py --seconds 100 --seed 601 2>&1 | tail -2
echo "GRB_MI355X_DETERMINISTIC=1:"
GRB_MI355X_DETERMINISTIC=1 timeout 150 python tools/fuzz_parity.py --seconds 60 --seed 602 2>&1 | tail -2
echo "GRB_MI355X_SELL=1 (the lane-per-piece layout wherever kernel X runs):"
GRB_MI355X_SELL=1 timeout 150 python tools/fuzz_parity.py --seconds 40 --seed 604 2>&1 | tail -2
timeout 150 python tools/fuzz_companions.py --seconds 50 --seed 603 2>&1 | tail -2
echo "tools/fuzz_batch.py (matrices of <= 64 very long rows as bitmaps: mxm against the oracle, element-wise chains against the generic kernels):"
timeout 200 python tools/fuzz_batch.py --seconds 90 --seed 605 2>&1 | tail -2
echo "GRB_MI355X_EWISE_FUSED=0 / GRB_MI355X_XT_NARROW=0 (the general element-wise route, the wide value plane):"
GRB_MI355X_EWISE_FUSED=0 GRB_MI355X_XT_NARROW=0 timeout 150 python tools/fuzz_companions.py --seconds 30 --seed 606 2>&1 | tail -2
} > $out/fuzz.log 2>&1
cat $out/fuzz.log
