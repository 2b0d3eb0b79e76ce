#!/bin/bash
# Places an UNTRACKED, git-ignored scratch copy of the reference package and its tests beside the repo so that one `gpurun`
# call can carry it to the GPU box (which has no /root/reference) for tools/ref_tests_gpu.sh and tools/ref_doctests_gpu.sh.
# Remove it again after the call (`rm -rf .refscratch`): the reference's sources are never part of this repository.
set -eu
cd "$(dirname "$0")/.."
rm -rf .refscratch && mkdir -p .refscratch/docs
cp -r /root/reference/pygraphblas /root/reference/tests .refscratch/
cp /root/reference/docs/test_mm.mm /root/reference/docs/test_tsvfile.tsv /root/reference/docs/test_binfile.grb .refscratch/docs/
cp -r /root/reference/demo .refscratch/demo        # the notebooks (tools/ref_notebooks_run.py), their font and data files
find .refscratch -name __pycache__ -prune -exec rm -rf {} +
echo "scratch copy in .refscratch/ (git-ignored); delete it after the gpurun call"
