#!/bin/bash
# round 4, session 7: where the boundary waves sit in the grid
TAG=${1:-r04s7}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for v in "GKOC_TUNE_8=100" "GKOC_TUNE_8=75" "GKOC_TUNE_8=50" "GKOC_TUNE_8=25" "GKOC_TUNE_8=0"; do
echo "-- $v"
for rep in 1 2; do
env $v GKO_SIM_ONLY=x timeout 300 python tools/dist_sim.py 256 8 3 600 2>&1 | grep "Distributed" | grep -v steps | tee -a $OUT/dist_sim.txt
done
env $v timeout 300 python tools/dist_sim.py 256 8 3 50 2>&1 | grep "one-kernel" | tee -a $OUT/dist_sim.txt
done
echo "== correctness with the waves in the middle"
GKOC_TUNE_8=50 timeout 900 python -m pytest tests/test_distributed.py -m gpu -q -x -k "one_kernel or fork or late or mirror or rccl" 2>&1 | tail -3
echo done
