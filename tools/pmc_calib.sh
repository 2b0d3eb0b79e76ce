#!/bin/bash
# What the memory-side counters report for known byte counts (tools/probes/fetch_calib) and, with the same counters, for the masked
# SpGEMM: L2 hits / misses, request sizes.  Counters in their own passes, no tracing flags.  usage: tools/pmc_calib.sh <outdir>
set -u
out=${1:-gpurun_out/pmc_calib}
case "$out" in /*) ;; *) out="$GRAFT_REPO_ROOT/$out";; esac
mkdir -p "$out"; cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
tools/probes/fetch_calib > "$out/calib_plain.txt" 2>&1
i=0
for p in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_BUBBLE_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $p --output-format csv -d "$out/c$i" -o pmc -- tools/probes/fetch_calib < /dev/null > "$out/c$i.log" 2>&1; echo "calib pass $i ($p) rc=$?" >> "$out/passes.log"
  timeout 120 rocprofv3 --pmc $p --output-format csv -d "$out/t$i" -o pmc -- python tools/tc_probe.py --scale 22 --reps 1 --serial < /dev/null > "$out/t$i.log" 2>&1; echo "tc pass $i rc=$?" >> "$out/passes.log"
done
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
with open(out + "/summary.txt", "w") as fo:
    for tag, title in (("c", "fetch_calib"), ("t", "triangle count (serial bins)")):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(int))
        for f in glob.glob(out + f"/{tag}[0-9]/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "")
                if not any(s in k for s in ("k_read", "k_rows4", "k_gather8", "spgemm", "k_spmv_tiles", "k_xp_merge", "k_xp_hot")): continue
                short = k.split("(")[0].replace("void grb::", "")[:80]
                agg[short][row["Counter_Name"]] += float(row["Counter_Value"]); calls[short][row["Counter_Name"]] += 1
        fo.write(f"== {title}\n")
        for k, cs in agg.items():
            fo.write(k + "\n")
            for c, v in sorted(cs.items()): fo.write(f"   {c:30s} launches={calls[k][c]:3d} per launch={v / calls[k][c]:.6g}\n")
print(open(out + "/summary.txt").read())
PY
cat "$out/calib_plain.txt" "$out/passes.log"
