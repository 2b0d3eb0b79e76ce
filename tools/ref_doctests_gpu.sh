#!/bin/bash
# The UNMODIFIED reference package's doctests (tests/test_doctest.py -> pygraphblas.run_doctests, pygraphblas/__init__.py:376-398)
# through shim/ on the GPU box: every example's printed output is the reference's own golden value.
# Needs the untracked scratch copy of /root/reference/{pygraphblas,tests} under .refscratch/ (see tools/ref_tests_gpu.sh).
# usage: tools/ref_doctests_gpu.sh <outdir>
set -u
out=${1:-gpurun_out/refdoctests}
case "$out" in /*) ;; *) out="$GRAFT_REPO_ROOT/$out";; esac
mkdir -p "$out"
# graphviz is not installed in the image: a recording stand-in (tools/ref_harness_stubs, harness only) lets the docstrings that
# draw their matrices go on to their arithmetic examples; docs/ holds the files three examples read by relative path
export PYTHONPATH="$GRAFT_REPO_ROOT/shim:$GRAFT_REPO_ROOT/.refscratch:$GRAFT_REPO_ROOT/tools/ref_harness_stubs" PYTHONDONTWRITEBYTECODE=1
mkdir -p "$GRAFT_REPO_ROOT/.refscratch/docs/imgs"     # draw_matrix examples save PNGs there
cd "$GRAFT_REPO_ROOT/.refscratch"
timeout 500 /opt/conda/bin/python3.9 - > "$out/doctests.log" 2>&1 <<'PY'
import doctest, faulthandler, sys, io
faulthandler.dump_traceback_later(450, exit=True, file=sys.__stderr__)
import pygraphblas as p
from pygraphblas import matrix, vector, descriptor, base, unaryop, binaryop, selectop
tot_f = tot_t = 0
for mod in (p, selectop, unaryop, binaryop, matrix, vector, descriptor, base):
    buf = io.StringIO(); old = sys.stdout; sys.stdout = buf
    try:
        r = doctest.testmod(mod, optionflags=doctest.ELLIPSIS, raise_on_error=False)
    finally:
        sys.stdout = old
    print(f"== {mod.__name__}: {r.attempted - r.failed} of {r.attempted} examples pass")
    tot_f += r.failed; tot_t += r.attempted
    txt = buf.getvalue()
    if txt: print(txt[-60000:])
print(f"TOTAL: {tot_t - tot_f} of {tot_t} doctest examples pass")
PY
echo "rc=$?" >> "$out/doctests.log"
grep -E "^== |^TOTAL|^rc=" "$out/doctests.log"
