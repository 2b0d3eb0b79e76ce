cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -x -q -k "narrow or other_types or sssp or shortest" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_mxv_vxm_gpu.py tests/test_subpanels_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/fuzz_parity.py --seconds 40 --seed 21 2>&1 | tail -1
timeout 300 python bench.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', d['ms_per_step'], d['roofline']['frac'])
o=d.get('sssp') or d['config'].get('sssp'); print('sssp', o['seconds'], o['ms_per_sweep'], o['roofline']['frac'], o['sweeps'], o.get('parity_vs_oracle'))
"
