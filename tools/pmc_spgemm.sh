#!/bin/bash
# Kernel stats and HBM-traffic counters of the masked SpGEMM (triangle count).  Counters in their own passes, no tracing flags,
# bins serialised on one stream (tools/tc_probe.py --serial).  usage: tools/pmc_spgemm.sh <outdir> <scale>
set -u
out=${1:-gpurun_out/pmc_tc}; scale=${2:-22}
case "$out" in /*) ;; *) out="$GRAFT_REPO_ROOT/$out";; esac
mkdir -p "$out"; cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o tc -- python tools/tc_probe.py --scale $scale --reps 3 > "$out/tc_line.json" 2> "$out/stats.err"
f=$(find "$out/stats" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats.csv"
find "$out/stats" -name '*kernel_trace.csv' -delete
i=0
for p in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $p --output-format csv -d "$out/p$i" -o pmc -- python tools/tc_probe.py --scale $scale --reps 1 --serial < /dev/null > "$out/p$i.log" 2>&1
  echo "pass $i rc=$?" >> "$out/passes.log"
done
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "spgemm" not in k and "k_bin_rows" not in k and "k_scatter_acc" not in k and "k_reduce" not in k: continue
        short = k.split("(")[0].replace("void grb::", "")[:90]
        agg[short][row["Counter_Name"]] += float(row["Counter_Value"]); calls[short][row["Counter_Name"]] += 1
with open(out + "/pmc_summary.txt", "w") as fo:
    tot = collections.defaultdict(float)
    for k, cs in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(cs.items()):
            fo.write(f"   {c:28s} launches={calls[k][c]:3d} sum={v:.6g}\n"); tot[c] += v
    fo.write("TOTAL over the kernels above (one triangle count)\n")
    for c, v in sorted(tot.items()): fo.write(f"   {c:28s} {v:.6g}\n")
    if "FETCH_SIZE" in tot: fo.write(f"   HBM read bytes  = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction) = {tot['FETCH_SIZE']*2048:.6g}\n")
    if "WRITE_SIZE" in tot: fo.write(f"   HBM write bytes = WRITE_SIZE KiB x 1024 = {tot['WRITE_SIZE']*1024:.6g}\n")
print(open(out + "/pmc_summary.txt").read()[-1500:])
PY
cat "$out/tc_line.json" | tail -1; cat "$out/passes.log"
