#!/bin/bash
# round-4 GPU session 2: where the time of the scale-25 pattern-only product goes: kernel trace + L2 / memory-side counters, S = 1 vs S = 8
out=gpurun_out/r4b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for S in 1 8; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_S$S -o kt -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels $S > $out/kt_S$S.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/p1_S$S -o pmc -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels $S > $out/p1_S$S.log 2>&1
  timeout 200 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d $out/p2_S$S -o pmc -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels $S > $out/p2_S$S.log 2>&1
done
python - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for S in (1, 8):
    print("==== S =", S)
    for f in glob.glob(f"{out}/kt_S{S}/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        for r in rows[:12]:
            print("  ", r["Name"].split("(")[0][-60:], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/p?_S{S}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "k_spmv_tiles" not in k and "k_xp_merge" not in k: continue
            agg[k.split("<")[0].replace("void grb::", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        print("  ", k)
        for c, v in sorted(cs.items()):
            print(f"      {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
rm -rf $out/*/*/*.db 2>/dev/null
find $out -name "*counter_collection.csv" -size +5M -delete
