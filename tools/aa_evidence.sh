out=gpurun_out/r05e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_mxm_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/aa_kt -o aa -- python tools/workloads.py --what aa --aa-methods hash > $out/aa_kt.log 2>&1
GRB_MI355X_DETERMINISTIC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/aa_kt_det -o aa -- python tools/workloads.py --what aa --aa-methods hash > $out/aa_kt_det.log 2>&1
python - $out <<'PY' > $out/aa_kernel_stats.txt
import csv, glob, sys
out = sys.argv[1]
for tag, title in (("aa_kt", "default mode"), ("aa_kt_det", "deterministic mode (GRB_MI355X_DETERMINISTIC=1)")):
    print(f"A @ A (unmasked GrB_mxm), symmetric R-MAT-18 FP64 PLUS_TIMES, two-pass hash path, {title}: rocprofv3 --kernel-trace --stats of tools/workloads.py --what aa --aa-methods hash (3 products)")
    for l in open(f"{out}/{tag}.log"):
        if l.startswith("{"): print("  ", l.strip()[:600])
    for f in glob.glob(f"{out}/{tag}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "grb::" in r["Name"] and float(r["TotalDurationNs"]) > 3e5: print(f'   {r["Name"].split("(")[0][-90:]:90s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us')
PY
cat $out/aa_kernel_stats.txt | cut -c1-170
find $out -name "*kernel_trace.csv" -delete
