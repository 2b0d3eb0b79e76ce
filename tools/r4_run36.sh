#!/bin/bash
out=gpurun_out/r4ar; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $out/pytest_tail.txt; cat $out/pytest_tail.txt
timeout 900 python bench.py > $out/bench_line.json 2> $out/bench_err.txt; tail -c 600 $out/bench_err.txt
python - $out <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_line.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d.get("first_call_ms"), d.get("second_call_ms"))
for k in ("mxm", "bfs", "pagerank", "pagerank_scale25", "sssp", "mxm_unmasked", "bc"):
    v = d.get(k) or {}
    print(k, {kk: v[kk] for kk in v if kk in ("seconds", "ms_per_iteration", "ms_per_sweep", "GTEPS", "ms", "roofline")} if v else None)
PY
