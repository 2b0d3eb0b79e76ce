#!/bin/bash
out=gpurun_out/r4j; mkdir -p $out
timeout 900 python tools/r4_subpanel_probe.py --skip-b --scale 24 --subpanels 1,2o,4o --reps 20 > $out/probeA24.log 2>&1; echo "rc=$?"
timeout 900 python tools/r4_subpanel_probe.py --skip-b --scale 23 --subpanels 1,2o --reps 20 > $out/probeA23.log 2>&1; echo "rc=$?"
grep -h '^{' $out/probeA24.log $out/probeA23.log | cut -c1-300
