#!/bin/bash
# round-4 evidence session (run on the GPU box through gpurun; tools/make_refscratch.sh first when the reference legs are wanted):
#   GPU tests, bench line, rocprofv3 kernel stats of the bench command, SpMV PMC passes, the loops' per-call timelines, the workloads line by
#   line, the scale-25 product's counters, and (with .refscratch/) the unmodified reference's tests / doctests / notebooks through shim/.
tag=${1:-r04}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
bash tools/gpu_round.sh $tag > $out/gpu_round.log 2>&1; tail -5 $out/gpu_round.log | cut -c1-200
PMC_PASSES=3 bash tools/pmc_spmv.sh $out/pmc_spmv --variants FP64.PLUS_TIMES --methods auto > $out/pmc_spmv.log 2>&1; tail -30 $out/pmc_spmv.log
timeout 300 python tools/bfs_probe.py > $out/bfs_per_call_timeline.txt 2>&1
timeout 300 python tools/sssp_probe.py > $out/sssp_per_call_timeline.txt 2>&1
timeout 900 python tools/workloads.py --what bfs,tc,pr,bc,bcfull,aa > $out/workloads_scale22.jsonl 2> $out/workloads.err; echo "workloads rc=$?"
# the unmasked product (A @ A, R-MAT-18): kernel times of the hash path + where its waves' cycles go
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/aa_kt -o aa -- python tools/workloads.py --what aa --aa-methods hash > $out/aa_kt.log 2>&1
bash tools/pmc_sq_aa.sh $out/aa_sq > $out/aa_sq.log 2>&1
python - $out <<'PY' > $out/aa_kernel_stats.txt
import csv, glob, sys
out = sys.argv[1]
print("A @ A (unmasked GrB_mxm), symmetric R-MAT-18 FP64 PLUS_TIMES, two-pass hash path: rocprofv3 --kernel-trace --stats of tools/workloads.py --what aa --aa-methods hash (3 products)")
for l in open(f"{out}/aa_kt.log"):
    if l.startswith("{"): print("  ", l.strip()[:700])
for f in glob.glob(f"{out}/aa_kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "grb::" in r["Name"] and float(r["TotalDurationNs"]) > 3e5: print(f'   {r["Name"].split("(")[0][-90:]:90s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us')
print("SQ counters (rocprofv3 --pmc, their own passes: tools/pmc_sq_aa.sh):")
try: print(open(f"{out}/aa_sq/sq_summary.txt").read())
except Exception as e: print("   (missing)", e)
PY
for S in 1 a; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/pr25_kt_$S -o kt -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels $S > $out/pr25_kt_$S.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/pr25_p1_$S -o pmc -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels $S > $out/pr25_p1_$S.log 2>&1
  timeout 200 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d $out/pr25_p2_$S -o pmc -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels $S > $out/pr25_p2_$S.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $out/pr25_p3_$S -o pmc -- python tools/r4_subpanel_probe.py --skip-a --pr-subpanels $S > $out/pr25_p3_$S.log 2>&1
done
python - $out <<'PY' > $out/pr25_summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for S in ("1", "a"):
    print("==== R-MAT-25 FP32 PageRank (gap/prmark.py loop), sub-panels:", "one table per XCD (S = 1)" if S == "1" else "the library's choice (S = 4, a table per sub-panel)")
    for l in open(f"{out}/pr25_kt_{S}.log"):
        if l.startswith("{") and '"S"' in l: print("  ", l.strip()[:300])
    for f in glob.glob(f"{out}/pr25_kt_{S}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Name"]
            if "grb::" in n and any(k in n for k in ("k_spmv_tiles", "k_xp_merge", "k_xp_hot", "k_vec_chain")):
                print(f'   {n.split("(")[0][-72:]:72s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us')
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/pr25_p?_{S}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "k_spmv_tiles" not in k and "k_xp_merge" not in k: continue
            agg[k.split("<")[0].replace("void grb::", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        print("  ", k, "(per launch)")
        for c, v in sorted(cs.items()):
            print(f"      {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
cat $out/pr25_summary.txt | head -60
if [ -d .refscratch ]; then
  bash tools/ref_tests_gpu.sh $out/reftests > $out/reftests.log 2>&1; tail -3 $out/reftests/pytest_reference.log
  bash tools/ref_doctests_gpu.sh $out/refdoctests > $out/refdoctests.log 2>&1; tail -3 $out/refdoctests.log
  bash tools/ref_notebooks_gpu.sh $out/refnotebooks > $out/refnotebooks.log 2>&1; tail -5 $out/refnotebooks.log
fi
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -size +2M -delete; find $out -name "*.db" -delete
du -sh $out
