#!/bin/bash
# Where the unmasked SpGEMM's wave cycles go: two SQ passes over tools/workloads.py --what aa (hash path).  usage: tools/pmc_sq_aa.sh <outdir>
set -u
out=${1:-gpurun_out/pmc_sq_aa}
mkdir -p "$out"; cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d "$out/p1" -o pmc -- python tools/workloads.py --scale 18 --what aa --aa-methods hash < /dev/null > "$out/p1.log" 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$out/p2" -o pmc -- python tools/workloads.py --scale 18 --what aa --aa-methods hash < /dev/null > "$out/p2.log" 2>&1
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_spgemm" not in k: continue
        agg[k.split("(")[0].replace("void grb::", "")[:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/sq_summary.txt", "w") as fo:
    for k, cs in agg.items():
        fo.write(k + "\n")
        wc = sum(cs.get("SQ_WAVE_CYCLES", [0])) / max(1, len(cs.get("SQ_WAVE_CYCLES", [1])))
        for c, v in sorted(cs.items()):
            m = sum(v) / len(v)
            fo.write(f"   {c:28s} n={len(v):3d} mean={m:.6g}" + (f"  ({100*m/wc:.1f}% of wave cycles)" if wc and c.startswith("SQ_") and "INSTS" not in c and "LDS_" not in c and "WAVES" not in c and "BUSY" not in c else "") + "\n")
print(open(out + "/sq_summary.txt").read())
PY
