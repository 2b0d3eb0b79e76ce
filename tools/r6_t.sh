cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_nonblocking_gpu.py tests/test_subpanels_gpu.py tests/test_reference_suite_gpu.py tests/test_shim.py tests/test_mxm_gpu.py -m gpu -x -q -k "compiled_chains or integer_division or same_bits or outside_the_pagerank or lane_per_piece or reference or shim or batch or betweenness" 2>&1 | tail -15
