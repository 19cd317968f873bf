#!/bin/bash
OUT=gpurun_out/r02s23; rm -rf $OUT; mkdir -p $OUT
bash tools/run_reftests.sh $OUT > /dev/null 2>&1
cat $OUT/summary.txt
timeout 300 python -m pytest tests/test_krylov_gpu.py -q -m gpu -x -k "adaptive or reduced" 2>&1 | tail -4
exit 0
