#!/bin/bash
# round-6 evidence session (run on the GPU box through gpurun; tools/make_refscratch.sh first when the reference legs are wanted):
#   GPU tests, bench line, rocprofv3 kernel stats of the bench command, SpMV and masked-SpGEMM PMC passes, the loops' per-call timelines, the
#   unmasked product's kernel stats at both sizes, the BC driver's kernel stats, the first BFS on a fresh matrix, and (with .refscratch/) the
#   unmodified reference's tests / doctests / notebooks through shim/.
tag=${1:-r06}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
bash tools/gpu_round.sh $tag --durations=10 > $out/gpu_round.log 2>&1; tail -5 $out/gpu_round.log | cut -c1-200
PMC_PASSES=2 bash tools/pmc_spmv.sh $out/pmc_spmv --variants FP64.PLUS_TIMES --methods auto > $out/pmc_spmv.log 2>&1; tail -24 $out/pmc_spmv.log
bash tools/pmc_spgemm.sh $out/pmc_tc 22 > $out/pmc_tc.log 2>&1; tail -12 $out/pmc_tc.log
timeout 300 python tools/bfs_probe.py > $out/bfs_per_call_timeline.txt 2>&1
timeout 300 python tools/sssp_probe.py > $out/sssp_per_call_timeline.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/aa_kt -o aa -- python tools/workloads.py --what aa --aa-methods hash > $out/aa_kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/aa_kt_wide -o aa -- python tools/workloads.py --what aa --aa-scale 20 --aa-edgefactor 4 --aa-methods hash > $out/aa_kt_wide.log 2>&1
GRB_MI355X_SPA_RANK=0 timeout 300 python tools/workloads.py --what aa --aa-scale 20 --aa-edgefactor 4 --aa-methods hash > $out/aa_wide_norank.log 2>&1
python - $out <<'PY' > $out/aa_kernel_stats.txt
import csv, glob, sys
out = sys.argv[1]
for tag, title in (("aa_kt", "symmetric R-MAT-18 (edge factor 16)"), ("aa_kt_wide", "symmetric R-MAT-20 (edge factor 4): 2^20 columns, ranked rows slab by slab, rows of few entries in one step")):
    print(f"A @ A (unmasked GrB_mxm), {title}, FP64 PLUS_TIMES, two-pass hash path: rocprofv3 --kernel-trace --stats of tools/workloads.py --what aa --aa-methods hash (3 products)")
    for l in open(f"{out}/{tag}.log"):
        if l.startswith("{"): print("  ", l.strip()[:700])
    for f in glob.glob(f"{out}/{tag}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "grb::" in r["Name"] and float(r["TotalDurationNs"]) > 3e5: print(f'   {r["Name"].split("(")[0][-90:]:90s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us')
print("the same wide product with GRB_MI355X_SPA_RANK=0 (block by block, as rounds 3-5 ran every result of more than 2^18 columns):")
for l in open(f"{out}/aa_wide_norank.log"):
    if l.startswith("{"): print("  ", l.strip()[:400])
PY
cat $out/aa_kernel_stats.txt | cut -c1-180
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bc_kt -o bc -- python tools/workloads.py --what bcfull > $out/bc_kt.log 2>&1
(echo "batched BC (gap/bcmark.py:16-67), R-MAT-22, ns = 4: rocprofv3 --kernel-trace --stats of tools/workloads.py --what bcfull (one first run + two timed runs in the process)"; grep "^{" $out/bc_kt.log | cut -c1-500; python tools/kstats.py $out/bc_kt 80 | grep "grb" | head -30) > $out/bc_kernel_stats.txt
GRB_MI355X_BATCH=0 timeout 300 python tools/workloads.py --what bcfull 2>/dev/null | tail -1 | cut -c1-400 >> $out/bc_kernel_stats.txt
head -12 $out/bc_kernel_stats.txt | cut -c1-200
timeout 600 python tools/workloads.py --what bfs,tc,pr,bc,bcfull,tcfp > $out/workloads_scale22.jsonl 2> $out/workloads.err; echo "workloads rc=$?"
if [ -d .refscratch ]; then
  bash tools/ref_tests_gpu.sh $out/reftests > $out/reftests.log 2>&1; tail -3 $out/reftests/pytest_reference.log
  bash tools/ref_doctests_gpu.sh $out/refdoctests > $out/refdoctests.log 2>&1; tail -3 $out/refdoctests.log
  bash tools/ref_notebooks_gpu.sh $out/refnotebooks > $out/refnotebooks.log 2>&1; tail -8 $out/refnotebooks.log
fi
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -size +2M -delete; find $out -name "*.db" -delete
du -sh $out
