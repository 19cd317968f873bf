#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02final
cd $GRAFT_REPO_ROOT
for sv in cg pipe_cg; do
  GKOC_EXAMPLE_TRACE=1 examples/native_dist_cg 16 3000 -1 $sv 8 mirror 2>&1 | grep -E "^\{|host us" | tee -a $OUT/native_dist_3000.txt
done
GKOC_EXAMPLE_TRACE=1 examples/native_dist_cg 16 3000 -1 cg 8 2>&1 | grep -E "^\{|host us" | tee -a $OUT/native_dist_3000.txt
GKOC_EXAMPLE_TRACE=1 examples/native_dist_cg 16 3000 -1 pipe_cg 8 2>&1 | grep -E "^\{|host us" | tee -a $OUT/native_dist_3000.txt
timeout 300 python tools/dist_host_cost.py 16 3000 direct 2>&1 | grep -E "^grid" | tee -a $OUT/native_dist_3000.txt
