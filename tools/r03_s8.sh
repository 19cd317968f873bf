#!/bin/bash
TAG=${1:-r03s8}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== arena role scenarios"
timeout 1200 python -m pytest tests/test_arena_roles_gpu.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids | tail -60 | tee $OUT/arena_roles.txt
echo "== dropin_bench (default arena)"
(cd oracle/_ref/dropin && GKOC_ARENA_VERBOSE=1 timeout 300 ./dropin_bench 256 30 50 2>&1 | grep -v "^\[gkoc arena\]   probe" | tail -14)
exit 0
