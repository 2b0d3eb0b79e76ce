cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
o=d.get('sssp') or d['config'].get('sssp'); print({k:o[k] for k in ('seconds','ms_per_sweep','last_sweep_plan','stored_value_bytes_note')}, o['roofline']['frac'])
"
