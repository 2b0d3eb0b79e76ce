cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_companion_ops_gpu.py tests/test_shim.py -m gpu -x -q 2>&1 | tail -8
bash tools/ref_notebooks_gpu.sh gpurun_out/r6nb > gpurun_out/r6nb.log 2>&1; grep -E "^== (Louvain|Centrality)|TOTAL|cell" gpurun_out/r6nb/notebooks.log | head -40
bash tools/ref_tests_gpu.sh gpurun_out/r6rt > gpurun_out/r6rt.log 2>&1; tail -2 gpurun_out/r6rt/pytest_reference.log
bash tools/ref_doctests_gpu.sh gpurun_out/r6dt > gpurun_out/r6dt.log 2>&1; tail -2 gpurun_out/r6dt.log
