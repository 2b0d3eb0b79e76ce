#!/bin/bash
# round 6: what the panel pipelines wait on — SQ, TA and TCP counters of k_spmv_tiles and k_spmv_sell (FP64 PLUS_TIMES, FP32 PLUS_SECOND, R-MAT-22),
# each group in a pass of its own (counters only: no trace domains).  usage: tools/r6_pmc_spmv.sh <outdir>
set -u
out=${1:-gpurun_out/r6pmc}; mkdir -p "$out"; cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
probe="python tools/sell_probe.py --scale 22 --cases FP64:PLUS_TIMES,FP32:PLUS_SECOND --variants tiles,sell --skip-b --reps 3"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" \
           "TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
           "FETCH_SIZE WRITE_SIZE TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -o pmc -- $probe < /dev/null > "$out/p$i.log" 2>&1 || echo "pass $i rc=$?"
done
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if not any(s in k for s in ("k_spmv_tiles", "k_spmv_sell", "k_xp_merge")): continue
        agg[k.split("(")[0].replace("void grb::", "")[:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, cs in sorted(agg.items()):
        fo.write(k + "\n")
        wc = sum(cs.get("SQ_WAVE_CYCLES", [0])) / max(1, len(cs.get("SQ_WAVE_CYCLES", [1])))
        for c, v in sorted(cs.items()):
            m = sum(v) / len(v)
            fo.write(f"   {c:36s} n={len(v):3d} mean={m:.6g}" + (f"  ({100*m/wc:.1f}% of wave cycles)" if wc and c.startswith("SQ_") and "INSTS" not in c and "LDS_" not in c and "BUSY" not in c else "") + "\n")
print(open(out + "/summary.txt").read())
PY
find "$out" -name "*.csv" -size +2M -delete
