#!/bin/bash
# One GPU-box session: GPU tests, bench line, rocprof kernel stats of the bench command.  usage: tools/gpu_round.sh <tag>
set -u
tag=${1:-r02}
out=gpurun_out/$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests -m gpu -x -q --timeout 300 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
tail -5 "$out/pytest_gpu.log"
timeout 300 python bench.py --steps 50 --warmup 5 > "$out/bench_line.json" 2> "$out/bench.err"; echo "bench rc=$?"
cat "$out/bench_line.json"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o bench -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline > "$out/bench_line_under_rocprof.json" 2> "$out/prof.err"
f=$(find "$out/prof" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/bench_kernel_stats.csv" && head -12 "$out/bench_kernel_stats.csv"
find "$out/prof" -name '*kernel_trace.csv' -delete
