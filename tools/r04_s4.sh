#!/bin/bash
# round 4, session 4: timelines of one rank's CG iteration (rocprofv3 kernel trace), variants
TAG=${1:-r04s4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"
timeout 900 python -m pytest tests/test_arena_classes_gpu.py tests/test_distributed.py -m gpu -q 2>&1 | tail -15 | tee $OUT/tests.txt
echo "== pieces"
timeout 300 python tools/dist_sim.py 256 8 3 400 2>&1 | grep -v "^rank\|amdgpu.ids" | tee $OUT/dist_sim_full.txt
for v in "default:" "nodot:GKO_GATED_DOT=0" "old:GKO_GATED_DOT=0 GKO_STEP1_CHECK=0 GKOC_COMM_FORK=event"; do
name=${v%%:*}; envs=${v#*:}
echo "== trace $name ($envs)"
cd /tmp
env $envs GKO_SIM_ONLY=cg rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -o t -- python $GRAFT_REPO_ROOT/tools/dist_sim.py 256 8 3 300 2>&1 | grep "DistributedCg"
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 36 | tee $OUT/timeline_$name.txt
done
echo done
