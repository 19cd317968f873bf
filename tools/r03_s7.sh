#!/bin/bash
TAG=${1:-r03s7}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== dist_sim under kernel trace (PipeCg fused only)"
(cd /tmp && GKO_SIM_ONLY=pipe timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ds -- python $GRAFT_REPO_ROOT/tools/dist_sim.py 256 8 3 200 > $OUT/trace_run.txt 2>&1)
grep -v amdgpu $OUT/trace_run.txt | tail -6
find $OUT/trace -name "*kernel_stats*" | head -1 | xargs -r -I{} cp {} $OUT/dist_sim_pipe_kernel_stats.csv
rm -rf $OUT/trace
python - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/dist_sim_pipe_kernel_stats.csv")))[:16]:
    print(r['Calls'], f"{float(r['AverageNs'])/1e3:9.1f} us  tot {float(r['TotalDurationNs'])/1e6:8.2f} ms", r['Name'][:110])
PY
echo "== dropin arena scenarios"
(cd oracle/_ref/dropin && timeout 300 ./dropin_bench 256 30 50 2>&1 | tail -8)
exit 0
