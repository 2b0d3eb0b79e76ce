cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
ls ~/.cache/grb_mi355x 2>/dev/null | wc -l
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -10
ls ~/.cache/grb_mi355x 2>/dev/null | wc -l
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -10
