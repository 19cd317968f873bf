#!/bin/bash
# round 4, session 3: the state of HEAD on a fresh box - arena classes, late gate, fresh-process
# bench lines, and the per-rank iteration with the criterion inside step_1 / <p,q> from the product
TAG=${1:-r04s3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== new tests"
timeout 900 python -m pytest tests/test_arena_classes_gpu.py tests/test_distributed.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/tests.txt
echo "== fresh bench x8"
for i in 1 2 3 4 5 6 7 8; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-ginkgo-api --no-pmc --cg-iters 30 > $OUT/bench_$i.json 2> $OUT/bench_$i.err
tail -1 $OUT/bench_$i.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['placement']; print(d['value'], d['config']['pct_hbm_peak'], d.get('cg_iters_per_s'), p['memory_classes_found'], p['class_of'], p['granules_walked'], p['granules_classified'], p['search_ms'], p['probe_retries'])"
done
echo "== dist_sim variants"
for v in "" "GKOC_COMM_FORK=event" "GKO_GATED_DOT=0" "GKO_STEP1_CHECK=0" "GKOC_TUNE_2=2" "GKO_GATED_DOT=0 GKO_STEP1_CHECK=0 GKOC_COMM_FORK=event"; do
echo "-- $v"
env $v GKO_SIM_ONLY=x timeout 300 python tools/dist_sim.py 256 8 3 2>&1 | grep -v "^rank" | tee -a $OUT/dist_sim.txt
done
echo done
