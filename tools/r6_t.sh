cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6pr; mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$out/prof" -o pr -- python tools/workloads.py --what pr > "$out/pr.json" 2> "$out/prof.err"
python tools/ktimeline.py "$out/prof" 40
find "$out/prof" -name '*kernel_trace.csv' -delete
