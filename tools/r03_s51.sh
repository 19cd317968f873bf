#!/bin/bash
OUT=gpurun_out/${1:-r03s51}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_distributed.py tests/test_native_cg_gpu.py -m gpu -x -q 2>&1 | tail -4
echo "== one-kernel product (gated)"; timeout 300 python tools/dist_sim.py 256 8 3 2>&1 | grep "fused=True  step_2 fused with its neighbour=True\|rror\|Trace\|whole\|one-kernel" | tee $OUT/dist_sim_gated.txt
echo "== side stream + join"; GKO_GATED_SPMV=0 timeout 300 python tools/dist_sim.py 256 8 3 2>&1 | grep "fused=True  step_2 fused with its neighbour=True\|whole" | tee $OUT/dist_sim_side.txt
