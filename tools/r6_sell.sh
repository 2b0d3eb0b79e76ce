#!/bin/bash
# round 6: the lane-per-piece layout experiment on the GPU box.  usage: tools/r6_sell.sh <tag> [probe args]
set -u
tag=${1:-r6sell}; shift || true
out=gpurun_out/$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GRB_MI355X_VERBOSE=1
timeout 600 python tools/sell_probe.py --scale 20 --pr-scale 20 --oracle --reps 20 > "$out/probe_s20.jsonl" 2> "$out/probe_s20.err"; echo "s20 rc=$?"
cut -c1-400 "$out/probe_s20.jsonl"; grep -v "stream\|xcd plan" "$out/probe_s20.err" | tail -5
timeout 900 python tools/sell_probe.py --scale 22 --pr-scale 22 "$@" > "$out/probe_s22.jsonl" 2> "$out/probe_s22.err"; echo "s22 rc=$?"
cut -c1-400 "$out/probe_s22.jsonl"; grep "lane-per-piece" "$out/probe_s22.err" | head -3; grep -v "stream\|xcd plan\|lane-per-piece" "$out/probe_s22.err" | tail -5
