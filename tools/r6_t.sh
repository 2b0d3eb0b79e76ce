cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_mxm_gpu.py -m gpu -x -q -k "batch or betweenness or few_long" 2>&1 | tail -3
for i in 1 2; do timeout 300 python tools/workloads.py --scale 22 --what bcfull 2>&1 | tail -1 | cut -c1-220; done
GRB_MI355X_SPMM=1 timeout 300 python tools/workloads.py --scale 22 --what bcfull 2>&1 | tail -1 | cut -c1-220
