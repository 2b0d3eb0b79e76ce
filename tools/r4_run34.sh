#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for i in 1 2; do timeout 300 python tools/r4_subpanel_probe.py --skip-a --pr-subpanels a 2>&1 | grep '^{' | tail -1 | cut -c1-160; done
timeout 900 python -m pytest tests/test_subpanels_gpu.py tests/test_mxv_vxm_gpu.py -x -q 2>&1 | tail -2
