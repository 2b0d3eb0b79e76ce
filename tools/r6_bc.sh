#!/bin/bash
# round 6: the batch matrices as bitmaps — parity tests, the per-step profile of the BC driver (twice: first run, then warm), whole-driver time
set -u
out=gpurun_out/${1:-r6bc}; mkdir -p "$out"; cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_mxm_gpu.py -m gpu -x -q -k "batch or betweenness or few_long or bc" 2>&1 | tail -15
for b in 0 1; do
  echo "== GRB_MI355X_BATCH=$b"
  GRB_MI355X_BATCH=$b timeout 300 python tools/bc_profile.py 22 > "$out/bc_profile_batch$b.txt" 2>&1; tail -45 "$out/bc_profile_batch$b.txt" | awk '{a[$1" "$2" "$3]+=$(NF-1); c[$1" "$2" "$3]++} END {for (k in a) printf "%-34s x%-3d %8.4f s\n", k, c[k], a[k]}' | sort
  GRB_MI355X_BATCH=$b timeout 300 python tools/workloads.py --scale 22 --what bcfull 2>&1 | tail -2 | cut -c1-600
done
