#!/bin/bash
# The UNMODIFIED reference package and its own tests, run on top of shim/ (CFFI -> libgrb_mi355x.so) on the GPU box.
# The reference tree is not part of this repository: the caller places an untracked scratch copy of
# /root/reference/{pygraphblas,tests} under .refscratch/ (git-ignored) before `gpurun` — tools/make_refscratch.sh — and
# deletes it after the call.
# usage: tools/ref_tests_gpu.sh <outdir> [pytest args...]
set -u
out=${1:-gpurun_out/reftests}; shift || true
case "$out" in /*) ;; *) out="$GRAFT_REPO_ROOT/$out";; esac
mkdir -p "$out"
cd "$GRAFT_REPO_ROOT"
PY=/opt/conda/bin/python3.9
export PYTHONPATH="$GRAFT_REPO_ROOT/shim:$GRAFT_REPO_ROOT/.refscratch"
# per-test watchdog (the conda python has no pytest-timeout): a conftest of the harness, placed beside the scratch copy of the tests
cat > "$GRAFT_REPO_ROOT/.refscratch/tests/conftest.py" <<'PYEOF'
import faulthandler, sys, pytest
@pytest.fixture(autouse=True)
def _watchdog():
    faulthandler.dump_traceback_later(60, exit=True, file=sys.__stderr__)
    yield
    faulthandler.cancel_dump_traceback_later()
PYEOF
cd /tmp
$PY -c "import pygraphblas as p; print('reference package imported from', p.__file__)" > "$out/import.log" 2>&1
timeout 400 $PY -m pytest -p no:faulthandler -c /dev/null --rootdir /tmp -p no:cacheprovider -q -rfE --tb=line "$@" \
  "$GRAFT_REPO_ROOT/.refscratch/tests/test_matrix.py" "$GRAFT_REPO_ROOT/.refscratch/tests/test_vector.py" \
  "$GRAFT_REPO_ROOT/.refscratch/tests/test_descriptor.py" "$GRAFT_REPO_ROOT/.refscratch/tests/test_scalar.py" \
  "$GRAFT_REPO_ROOT/.refscratch/tests/test_base.py" "$GRAFT_REPO_ROOT/.refscratch/tests/test_types.py" > "$out/pytest_reference.log" 2>&1
echo "rc=$?" >> "$out/pytest_reference.log"
cat "$out/import.log"; tail -60 "$out/pytest_reference.log"
