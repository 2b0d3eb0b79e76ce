#!/bin/bash
set -u
out=gpurun_out/${1:-r6bcprof}; mkdir -p "$out"; cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_mxm_gpu.py -m gpu -x -q -k "batch or betweenness" 2>&1 | tail -4
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o bc -- python tools/workloads.py --scale 22 --what bcfull > "$out/bcfull.json" 2> "$out/prof.err"
tail -1 "$out/bcfull.json" | cut -c1-400
python tools/kstats.py "$out/prof" 30 | tee "$out/bc_kernel_stats.txt"
find "$out/prof" -name '*kernel_trace.csv' -delete
