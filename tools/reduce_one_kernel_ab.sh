#!/bin/bash
# A/B of GKOC_TUNE_REDUCE_ONE_KERNEL (key 9): its tests, one rank's iteration of 256^3 / 8, the single-GPU CG
TAG=${1:-r04_reduce_ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"
timeout 300 python -m pytest tests/test_reduce_one_kernel_gpu.py -q -x 2>&1 | tail -5 | tee $OUT/tests.txt
for v in "GKOC_TUNE_9=0" "GKOC_TUNE_9=1" "GKOC_TUNE_9=0" "GKOC_TUNE_9=1"; do
echo "-- $v"
env $v GKO_SIM_ONLY=x timeout 300 python tools/dist_sim.py 256 8 3 600 2>&1 | grep "Distributed" | tee -a $OUT/dist_sim.txt
done
echo done
