#!/bin/bash
TAG=${1:-r03s6}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== distributed tests"
timeout 1500 python -m pytest tests/test_distributed.py -m gpu -q -x 2>&1 | tail -12
echo "== dist_sim 256 8 3"
timeout 600 python tools/dist_sim.py 256 8 3 200 2>&1 | grep -v amdgpu.ids | tee $OUT/dist_sim_256_8.txt | tail -18
exit 0
echo "== dist_sim 512 8 3"
timeout 600 python tools/dist_sim.py 512 8 3 100 2>&1 | grep -v amdgpu.ids | tee $OUT/dist_sim_512_8.txt | tail -18
exit 0
