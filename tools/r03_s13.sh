#!/bin/bash
TAG=${1:-r03s13}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== distributed tests"
timeout 1500 python -m pytest tests/test_distributed.py -m gpu -q -x 2>&1 | tail -12
exit 0
echo "== dist_sim 256 8 3 (reduction + exchange behind one fork)"
GKO_SIM_ONLY=x timeout 600 python tools/dist_sim.py 256 8 3 200 2>&1 | grep -v amdgpu.ids | tee $OUT/dist_sim_joint.txt | tail -6
echo "== dist_sim 256 8 3, GKO_SIM_SEPARATE_REDUCE=1 (all_reduce_begin / _end with their own events)"
GKO_SIM_SEPARATE_REDUCE=1 GKO_SIM_ONLY=x timeout 600 python tools/dist_sim.py 256 8 3 200 2>&1 | grep -v amdgpu.ids | tee $OUT/dist_sim_separate.txt | tail -6
exit 0
