#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03s52}
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(cd /tmp && GKO_SIM_ONLY=cg timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o ds -- python $GRAFT_REPO_ROOT/tools/dist_sim.py 256 8 3 100 > $OUT/trace_run.txt 2>&1)
grep "Distributed" $OUT/trace_run.txt
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 30 | tee $OUT/timeline_gated_cg.txt
rm -rf $OUT/trace
