#!/bin/bash
# One GPU-box session: GPU tests, bench line, rocprof kernel stats of the bench command.  usage: tools/gpu_round.sh <tag> [pytest args]
set -u
tag=${1:-r03}; shift || true
out=gpurun_out/$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 600 --durations=8 "$@" > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
tail -14 "$out/pytest_gpu.log"
timeout 400 python bench.py > "$out/bench_line.json" 2> "$out/bench.err"; echo "bench rc=$?"; tail -3 "$out/bench.err"
cat "$out/bench_line.json"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o bench -- python bench.py --no-cpu-baseline > "$out/bench_line_under_rocprof.json" 2> "$out/prof.err"
f=$(find "$out/prof" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/bench_kernel_stats.csv" && head -12 "$out/bench_kernel_stats.csv" | cut -c1-160
find "$out/prof" -name '*kernel_trace.csv' -delete
