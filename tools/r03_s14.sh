#!/bin/bash
TAG=${1:-r03s14}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== distributed + native tests"
timeout 1500 python -m pytest tests/test_distributed.py tests/test_native_cg_gpu.py -m gpu -q -x 2>&1 | tail -12
echo "== dist_sim 256 8 3 (high-priority side stream)"
GKO_SIM_ONLY=x timeout 600 python tools/dist_sim.py 256 8 3 200 2>&1 | grep -v amdgpu.ids | tee $OUT/dist_sim.txt | tail -6
echo "== native driver host cost (mirror, 16^3, 3000 iterations)"
(cd examples && for m in cg pipe_cg; do timeout 300 ./native_dist_cg 16 3000 1e-300 $m 4 mirror 2>&1 | tail -4; done)
exit 0
