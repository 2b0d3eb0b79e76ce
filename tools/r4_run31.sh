#!/bin/bash
out=gpurun_out/r4an; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_nonblocking_gpu.py -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_baseline_configs_gpu.py -x -q -k "pagerank" 2>&1 | tail -3
timeout 300 python tools/r4_subpanel_probe.py --skip-a --pr-subpanels a 2>&1 | grep '^{' | cut -c1-200
GRB_MI355X_LAZY_STORE_REDUCED=1 timeout 300 python tools/r4_subpanel_probe.py --skip-a --pr-subpanels a 2>&1 | grep '^{' | cut -c1-200
