#!/bin/bash
# Measurement builds of the masked SpGEMM (make XTFLAGS=-DSPG_EXP=n, see grb_spgemm_kernels.hpp) against the product: whole triangle
# count, concurrent bins and serial bins, and per-kernel times of the serial run.  usage: tools/spg_variants.sh <outdir> <lib>...
set -u
out=$1; shift
case "$out" in /*) ;; *) out="$GRAFT_REPO_ROOT/$out";; esac
mkdir -p "$out"; cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  export GRB_MI355X_LIB="$GRAFT_REPO_ROOT/pygraphblas_amd/$lib"
  echo "== $tag" | tee -a "$out/summary.txt"
  python tools/tc_probe.py --scale 22 --reps 4 2>/dev/null | tee -a "$out/summary.txt"
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag" -o tc -- python tools/tc_probe.py --scale 22 --reps 2 --serial > "$out/$tag.serial.json" 2>/dev/null
  cat "$out/$tag.serial.json" | tee -a "$out/summary.txt"
  f=$(find "$out/$tag" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a "$out/summary.txt"
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "spgemm" in n or "k_bin_rows" in n or "k_scatter_acc" in n or "k_reduce" in n or "k_flags" in n:
        print("   %-100s calls %3s avg %8.3f ms" % (n.replace("void grb::", "")[:100], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
  find "$out/$tag" -name '*kernel_trace.csv' -delete
done
