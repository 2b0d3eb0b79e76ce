cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -x -q -k "batched_bc" 2>&1 | tail -8
timeout 300 python tools/fuzz_companions.py --seconds 60 2>&1 | tail -3
