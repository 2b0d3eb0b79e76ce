#!/bin/bash
# PMC passes for the SpMV kernel (run on the GPU box).  Counters are collected in separate runs
# without any tracing flags, as the MI355X guide prescribes (TCC has 4 slots; FETCH_SIZE needs 3).
# usage: tools/pmc_spmv.sh <outdir> [probe args...]      (the numbers in profiles/: --variants FP64.PLUS_TIMES --methods auto; PMC_PASSES=2 = traffic only)
set -u
out=${1:-gpurun_out/pmc}; shift || true
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
passes=(
 "FETCH_SIZE"
 "WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum"
 "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum"
 "SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"
)
i=0
for p in "${passes[@]}"; do
  i=$((i+1))
  if [ -n "${PMC_PASSES:-}" ] && [ "$i" -gt "$PMC_PASSES" ]; then break; fi       # PMC_PASSES=2: the two traffic passes only
  timeout 150 rocprofv3 --pmc $p --output-format csv -d "$out/p$i" -o pmc -- python tools/spmv_probe.py --reps 3 "$@" < /dev/null > "$out/p$i.log" 2>&1
done
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_spm" not in k and "k_xp_" not in k and "k_wp_permute" not in k: continue
        short = k.split("(")[0].replace("void grb::", "")
        agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, cs in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(cs.items()):
            fo.write(f"   {c:40s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open(out + "/summary.txt").read())
PY
