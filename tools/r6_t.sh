cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_mxv_vxm_gpu.py -m gpu -x -q -k "code_bytes" 2>&1 | tail -12
