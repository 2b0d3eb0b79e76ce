set -u
out=gpurun_out/r02final3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py --steps 50 --warmup 5 > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_line_under_rocprof.json 2> $out/prof.err
f=$(find $out/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/bench_kernel_stats.csv
find $out/prof -name '*kernel_trace.csv' -delete
tail -c 400 $out/bench_line.json
