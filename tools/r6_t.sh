cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/threads_probe.py --scale 18 --rounds 5 2>&1 | grep -v amdgpu | tail -12
timeout 600 python -m pytest tests/test_mxv_vxm_gpu.py tests/test_nonblocking_gpu.py -m gpu -x -q 2>&1 | tail -3
