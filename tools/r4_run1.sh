#!/bin/bash
# round-4 GPU session 1: sub-panel plans (scale 22 / 25), the library exchange path through the RCCL stand-in, the SpMV parity tests
mkdir -p gpurun_out/r4a
export GRB_MI355X_VERBOSE=1
GRB_MI355X_XC_VERIFY=1 timeout 600 python tools/r4_subpanel_probe.py --oracle --skip-b > gpurun_out/r4a/probeA.log 2>&1; echo "probeA rc=$?"
timeout 900 python tools/r4_subpanel_probe.py --skip-a > gpurun_out/r4a/probeB.log 2>&1; echo "probeB rc=$?"
unset GRB_MI355X_VERBOSE
timeout 900 python -m pytest tests/test_dist_gpu.py -x -q > gpurun_out/r4a/dist.log 2>&1; echo "dist rc=$?"
timeout 900 python -m pytest tests/test_mxv_vxm_gpu.py tests/test_baseline_configs_gpu.py -x -q > gpurun_out/r4a/mxv.log 2>&1; echo "mxv rc=$?"
grep -h '^{' gpurun_out/r4a/probeA.log gpurun_out/r4a/probeB.log | cut -c1-330
tail -5 gpurun_out/r4a/dist.log gpurun_out/r4a/mxv.log
