#!/bin/bash
# round 6, session 14: counters for the CSR kernel on 81-entry rows next to SELL-P on the same data (what does the
# row-segment kernel wait for that SELL-P does not?), and for float values
OUT=gpurun_out/r06s14
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/pmc_groups.sh r06s14/flan 'csr_spmv_pipe3|sellp_spmv' -- python $GRAFT_REPO_ROOT/tools/flan_pmc.py 80 6 > $OUT/flan_pmc.log 2>&1
sed -n 1,12p $OUT/flan/summary.txt
rm -rf $OUT/flan/trace $OUT/flan/pmc_*/*/*.db
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
