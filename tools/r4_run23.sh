#!/bin/bash
out=gpurun_out/r4x; mkdir -p $out
timeout 900 python -m pytest tests/test_companion_ops_gpu.py tests/test_mxm_gpu.py tests/test_reference_suite_gpu.py "tests/test_baseline_configs_gpu.py::test_batched_bc_rmat22_against_the_oracle" -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $out/tests.log | grep -v "Librccl\|Hostname"
timeout 600 python tools/workloads.py --what bcfull 2>/dev/null | cut -c1-330
