#!/bin/bash
# quick GPU check: the baseline-config parity tests, then the bench line and the rocprof kernel stats of the bench command
set -u
tag=${1:-q}; out=gpurun_out/$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o bench -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline > "$out/bench_line_under_rocprof.json" 2> "$out/prof.err"
python - "$out" <<'PY'
import csv,sys,glob,json
out=sys.argv[1]
f=glob.glob(out+"/prof/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'grb::' in n and int(r['Calls'])>50: print(n.replace('void ','')[:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
d=json.loads(open(out+"/bench_line_under_rocprof.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
PY
find "$out/prof" -name '*kernel_trace.csv' -delete
