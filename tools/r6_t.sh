cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -8
timeout 200 python tools/fuzz_parity.py --seconds 40 --seed 31 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
o=d.get('bfs') or d['config'].get('bfs'); print('bfs', o['seconds'], o['roofline']['frac'], o.get('seconds_runs'), o.get('plans', o.get('kernels'))[:3] if (o.get('plans') or o.get('kernels')) else '')
"
GRB_MI355X_CODE_BYTES=0 timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-100
GRB_MI355X_CODE_BYTES=0 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
o=d.get('bfs') or d['config'].get('bfs'); print('bfs without code bytes', o['seconds'], o['roofline']['frac'], o.get('seconds_runs'))
"
