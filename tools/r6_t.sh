cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_companion_ops_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q -k "iseq or sssp or shortest" 2>&1 | tail -3
timeout 300 python tools/sssp_probe.py --only-async 2>&1 | grep -v amdgpu | tail -20
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
o=d.get('sssp') or d['config'].get('sssp'); print('sssp', o['seconds'], o['ms_per_sweep'], o['roofline']['frac'], o.get('parity_vs_oracle'))
"
