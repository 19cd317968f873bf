#!/bin/bash
TAG=${1:-r03s9}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== dist_sim 256 8 3 (complete boundary rows, v3 kernel)"
timeout 600 python tools/dist_sim.py 256 8 3 200 2>&1 | grep -v amdgpu.ids | tee $OUT/dist_sim_full.txt | tail -16
echo "== dist_sim 256 8 3, GKO_FULL_BOUNDARY=0"
GKO_FULL_BOUNDARY=0 GKO_SIM_ONLY=x timeout 600 python tools/dist_sim.py 256 8 3 200 2>&1 | grep -v amdgpu.ids | tee $OUT/dist_sim_old.txt | tail -8
for which in pipe cg; do
echo "== timeline $which"
(cd /tmp && GKO_SIM_ONLY=$which timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$which -o ds -- python $GRAFT_REPO_ROOT/tools/dist_sim.py 256 8 3 60 > $OUT/trace_run_$which.txt 2>&1)
f=$(find $OUT/trace_$which -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 36 | tee $OUT/timeline_$which.txt
rm -rf $OUT/trace_$which
done
exit 0
