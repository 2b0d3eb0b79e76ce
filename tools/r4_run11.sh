#!/bin/bash
out=gpurun_out/r4k; mkdir -p $out
GRB_MI355X_VERBOSE=1 timeout 900 python tools/r4_subpanel_probe.py --skip-a --pr-subpanels a > $out/probeB.log 2>&1; echo "rc=$?"
grep -h "tables serve" $out/probeB.log | head -3
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
grep -h '^{' $out/probeB.log | cut -c1-300
tail -n 3 $out/tests.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4k/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d.get("parity_vs_oracle"))
e = d.get("spmv_extra", {}); print({k: e.get(k) for k in ("first_call_ms", "second_call_ms", "frac_without_plan", "frac_rowblock", "plan_build_ms")})
for k in ("mxm", "bfs", "pagerank", "pagerank_scale25", "sssp"):
    if k in d: print(k, {kk: vv for kk, vv in d[k].items() if kk in ("seconds", "ms_per_iteration", "GTEPS", "GFLOPS", "ms_per_sweep", "parity_vs_oracle", "kernel")}, d[k].get("roofline", {}).get("frac"), d[k].get("cpu_baseline", {}).get("value"))
PY
