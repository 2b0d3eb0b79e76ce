#!/bin/bash
# The reference's demo notebooks (demo/*.ipynb), code cells executed unmodified through shim/ on the GPU box
# (tools/ref_notebooks_run.py).  Needs tools/make_refscratch.sh before the gpurun call.   usage: tools/ref_notebooks_gpu.sh <outdir>
set -u
out=${1:-gpurun_out/refnotebooks}
case "$out" in /*) ;; *) out="$GRAFT_REPO_ROOT/$out";; esac
mkdir -p "$out"
export PYTHONPATH="$GRAFT_REPO_ROOT/shim:$GRAFT_REPO_ROOT/.refscratch:$GRAFT_REPO_ROOT/tools/ref_harness_stubs" PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg
timeout 900 /opt/conda/bin/python3.9 "$GRAFT_REPO_ROOT/tools/ref_notebooks_run.py" "$GRAFT_REPO_ROOT/.refscratch/demo" > "$out/notebooks.log" 2>&1
echo "rc=$?" >> "$out/notebooks.log"
grep -E "^== |^TOTAL|^rc=" "$out/notebooks.log"
