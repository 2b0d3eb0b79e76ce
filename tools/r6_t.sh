cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_mxm_gpu.py -m gpu -x -q -k "every_row_kind" 2>&1 | tail -15
timeout 400 python tools/fuzz_batch.py --seconds 240 --seed 1 2>&1 | tail -12
