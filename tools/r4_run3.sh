#!/bin/bash
# round-4 GPU session 3: plan policy (kernel W first, sampled ranking), parity tests, a bench line
out=gpurun_out/r4c; mkdir -p $out
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py tests/test_nonblocking_gpu.py tests/test_mxv_vxm_gpu.py -x -q > $out/tests.log 2>&1; echo "tests rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
tail -4 $out/tests.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4c/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_blocks")}, d["roofline"]["frac"])
print(json.dumps(d.get("spmv_extra"), indent=1)[:3000])
for k in ("mxm", "bfs", "pagerank", "pagerank_scale25", "sssp"):
    if k in d: print(k, {kk: vv for kk, vv in d[k].items() if kk in ("seconds", "ms", "ms_per_iteration", "GTEPS", "GFLOPS", "ms_per_sweep", "parity_vs_oracle")}, d[k].get("roofline", {}).get("frac"))
PY
