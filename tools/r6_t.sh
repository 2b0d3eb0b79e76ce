cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for lib in .ab/libgrb_r5.so pygraphblas_amd/libgrb_mi355x.so; do
    echo "== $lib"
    GRB_MI355X_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python tools/bfs_probe.py --only-async 2>&1 | grep -E "total"
    GRB_MI355X_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python tools/sssp_probe.py --only-async 2>&1 | grep -E "total"
  done
done
for lib in .ab/libgrb_r5.so pygraphblas_amd/libgrb_mi355x.so; do
  echo "== $lib"
  GRB_MI355X_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/workloads.py --what bfs,pr 2>/dev/null | cut -c1-330
done
