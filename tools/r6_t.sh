cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/call_overhead_probe.py 2>&1 | grep -v amdgpu | tail -4
