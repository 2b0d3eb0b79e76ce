cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_companion_ops_gpu.py tests/test_vector.py tests/test_mxv_vxm_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 200 python tools/call_overhead_probe.py 2>&1 | grep -E "assign_scalar|r\[:\]|r.assign" 
for i in 1 2 3; do timeout 200 python tools/bfs_probe.py --only-async 2>&1 | grep total; done
