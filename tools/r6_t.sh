cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_mxm_gpu.py -m gpu -x -q -k "unmasked or hash or deterministic_mode_of_the_unmasked" 2>&1 | tail -6
timeout 300 python tools/workloads.py --what aa --aa-scale 20 --aa-edgefactor 4 --aa-methods hash 2>&1 | tail -1 | cut -c1-420
GRB_MI355X_DETERMINISTIC=1 timeout 300 python tools/workloads.py --what aa --aa-scale 20 --aa-edgefactor 4 --aa-methods hash 2>&1 | tail -1 | cut -c1-300
out=gpurun_out/r6aa2; mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o aa -- python tools/workloads.py --what aa --aa-scale 20 --aa-edgefactor 4 --aa-methods hash > "$out/aa.json" 2> "$out/prof.err"
python tools/kstats.py "$out/prof" 40 2>/dev/null | grep "grb" | head -4
find "$out/prof" -name '*kernel_trace.csv' -delete
