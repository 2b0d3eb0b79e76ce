#!/bin/bash
out=gpurun_out/r4ab; mkdir -p $out
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench$i.json 2> $out/bench$i.err
  python - $out/bench$i.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["ms_per_step"], {k: (d[k].get("ms_per_iteration"), d[k]["roofline"]["frac"], d[k].get("kernel", "")[:70]) for k in ("pagerank", "pagerank_scale25")})
PY
done
