cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -6
timeout 200 python tools/fuzz_parity.py --seconds 60 --seed 11 2>&1 | tail -2
timeout 200 python tools/fuzz_companions.py --seconds 40 --seed 5 2>&1 | tail -2 | cut -c1-200
