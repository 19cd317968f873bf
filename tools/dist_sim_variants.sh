#!/bin/bash
# round 4, session 5: fork by the product's first wave, fence only for waves that waited, <p,q> with its
# b entries prefetched and a wide fold
TAG=${1:-r04s5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"
timeout 900 python -m pytest tests/test_distributed.py -m gpu -q -x 2>&1 | tail -15 | tee $OUT/tests.txt
for v in "" "GKO_STEP_GATE=0" "GKO_DEFERRED_FORK=0" "GKO_GATED_DOT=1"; do
echo "-- $v"
for rep in 1 2; do
env $v GKO_SIM_ONLY=x timeout 300 python tools/dist_sim.py 256 8 3 600 2>&1 | grep "Distributed" | tee -a $OUT/dist_sim.txt
done
done
echo "== pieces"
timeout 300 python tools/dist_sim.py 256 8 3 100 2>&1 | grep "^   " | tee $OUT/pieces.txt
for v in "default:" ; do
name=${v%%:*}; envs=${v#*:}
echo "== trace $name ($envs)"
cd /tmp
env $envs GKO_SIM_ONLY=x rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -o t -- python $GRAFT_REPO_ROOT/tools/dist_sim.py 256 8 3 300 2>&1 | grep "Distributed"
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 60 | tee $OUT/timeline_$name.txt
done
echo done
