cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
 GRB_MI355X_EWISE_FUSED=0 timeout 300 python tools/workloads.py --what bcfull 2>/dev/null | tail -1 | cut -c100-180
 timeout 300 python tools/workloads.py --what bcfull 2>/dev/null | tail -1 | cut -c100-180
done
out=gpurun_out/r6bc3; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bc_kt -o bc -- python tools/workloads.py --what bcfull > $out/bc_kt.log 2>&1
python tools/kstats.py $out/bc_kt 80 | grep "grb" | head -24
find $out -name '*kernel_trace.csv' -delete
