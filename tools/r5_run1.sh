#!/bin/bash
# round 5, GPU session 1: full GPU suite, BFS timeline, bench line
set -u
out=gpurun_out/${1:-r5a}; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 --durations=6 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
tail -12 "$out/pytest_gpu.log"
timeout 120 python tools/bfs_probe.py > "$out/bfs_timeline.txt" 2>&1; cat "$out/bfs_timeline.txt" | head -40
timeout 500 python bench.py > "$out/bench_line.json" 2> "$out/bench.err"; echo "bench rc=$?"; tail -3 "$out/bench.err"
python - "$out" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+"/bench_line.json").read().strip().splitlines()[-1])
print("spmv", d["ms_per_step"], d["roofline"]["frac"])
for k in ("mxm","bfs","pagerank","pagerank_scale25","sssp","aa","bc"):
    o=d.get(k,{}); print(k, o.get("seconds", o.get("ms_per_iteration")), o.get("roofline",{}).get("frac"), o.get("parity_vs_oracle"), (o.get("cpu_baseline") or {}))
PY
