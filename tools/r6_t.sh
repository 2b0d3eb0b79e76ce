cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6bfs; mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$out/prof" -o bfs -- python tools/bfs_probe.py --only-async > "$out/bfs.txt" 2> "$out/prof.err"
grep -v amdgpu.ids $out/bfs.txt | head -30
python tools/ktimeline.py "$out/prof" 45
find "$out/prof" -name '*kernel_trace.csv' -delete
