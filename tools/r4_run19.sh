#!/bin/bash
out=gpurun_out/r4s; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
tail -n 3 $out/tests.log | grep -v "Librccl\|Hostname"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4s/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d.get("parity_vs_oracle"))
e = d.get("spmv_extra", {}); print({k: e.get(k) for k in ("first_call_ms", "second_call_ms", "frac_without_plan", "frac_rowblock", "plan_build_ms")}, e.get("permuted_labels", {}).get("frac"))
for k in ("mxm", "bfs", "pagerank", "pagerank_scale25", "sssp"):
    if k in d: print(k, {kk: vv for kk, vv in d[k].items() if kk in ("seconds", "ms_per_iteration", "GTEPS", "GFLOPS", "ms_per_sweep", "parity_vs_oracle")}, d[k].get("roofline", {}).get("frac"))
PY
