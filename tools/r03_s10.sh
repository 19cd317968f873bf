#!/bin/bash
TAG=${1:-r03s10}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== distributed + native tests"
timeout 1500 python -m pytest tests/test_distributed.py tests/test_native_cg_gpu.py -m gpu -q -x 2>&1 | tail -8
echo "== dist_sim 256 8 3"
timeout 600 python tools/dist_sim.py 256 8 3 200 2>&1 | grep -v amdgpu.ids | tee $OUT/dist_sim_256_8.txt | tail -18
echo "== dist_sim 256 8 3, GKO_FULL_BOUNDARY=0"
GKO_FULL_BOUNDARY=0 GKO_SIM_ONLY=x timeout 600 python tools/dist_sim.py 256 8 3 200 2>&1 | grep -v amdgpu.ids | tee $OUT/dist_sim_old.txt | tail -5
exit 0
