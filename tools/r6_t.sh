cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6aa; mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o aa -- python tools/workloads.py --what aa --aa-scale 20 --aa-edgefactor 4 --aa-methods hash > "$out/aa.json" 2> "$out/prof.err"
python tools/kstats.py "$out/prof" 40 2>/dev/null | grep -v "at::\|rocclr" | head -14
find "$out/prof" -name '*kernel_trace.csv' -delete
