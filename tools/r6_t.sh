cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_mxm_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q -k "batch or bc" 2>&1 | tail -5
timeout 300 python tools/fuzz_batch.py --seconds 60 --seed 3 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python tools/workloads.py --what bcfull 2>/dev/null | tail -1 | cut -c1-200; done
out=gpurun_out/r6bc2; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bc_kt -o bc -- python tools/workloads.py --what bcfull > $out/bc_kt.log 2>&1
python tools/kstats.py $out/bc_kt 80 | grep "spb" | head
find $out -name '*kernel_trace.csv' -delete
