#!/bin/bash
out=gpurun_out/r4g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
S=4o
timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/p1 -o pmc -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels $S > $out/p1.log 2>&1
timeout 200 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d $out/p2 -o pmc -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels $S > $out/p2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $out/p3 -o pmc -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels $S > $out/p3.log 2>&1
python - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p?/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_spmv_tiles" not in k and "k_xp_merge" not in k: continue
        agg[k.split("<")[0].replace("void grb::", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    print("  ", k)
    for c, v in sorted(cs.items()):
        print(f"      {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
find $out -name "*counter_collection.csv" -size +5M -delete
