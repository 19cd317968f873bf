#!/bin/bash
# round 4, session 8: the headline on a box we do not control - 20 fresh processes, the state that
# broke round 3's search (one class owns the small free blocks), forced 2-class outcome with rocprof
TAG=${1:-r04s8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
line() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['placement']; print(d['value'], d['config']['pct_hbm_peak'], d.get('cg_iters_per_s'), 'classes', p['memory_classes_found'], p['class_of'], 'walked', p['granules_walked'], 'classified', p['granules_classified'], 'search_ms', p['search_ms'], 'retries', p['probe_retries'])"; }
echo "== 20 fresh processes: python bench.py --gpus 1 --steps 20 --warmup 5 (no CPU / Ginkgo-API / PMC legs)"
for i in $(seq 1 20); do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-ginkgo-api --no-pmc --cg-iters 30 > $OUT/fresh_$i.json 2> $OUT/fresh_$i.err
line $OUT/fresh_$i.json
done | tee $OUT/fresh20.txt
echo "== survey of a fresh device (tools/class_lab survey 24)"
timeout 300 tools/class_lab survey 24 2>&1 | tee $OUT/survey_fresh.txt | tail -40
echo "== the state of BENCH_r03's box: another process holds the small free blocks of two classes (tools/class_lab starve)"
tools/class_lab starve 140 > $OUT/starve.txt 2>&1 &
HOLDER=$!
for i in $(seq 1 180); do grep -q READY $OUT/starve.txt && break; sleep 1; done
tail -5 $OUT/starve.txt
for i in 1 2 3; do
GKOC_ARENA_VERBOSE=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-ginkgo-api --no-pmc --cg-iters 30 > $OUT/starved_$i.json 2> $OUT/starved_$i.err
line $OUT/starved_$i.json
grep "survey" $OUT/starved_$i.err
done | tee $OUT/starved.txt
kill $HOLDER; wait $HOLDER 2>/dev/null
echo "== forced 2-class outcome under rocprofv3 --kernel-trace --stats"
cd /tmp
GKOC_ARENA_MAX_CLASSES=2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o two -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-ginkgo-api --no-pmc --cg-iters 0 > $GRAFT_REPO_ROOT/$OUT/two_classes.json 2> $GRAFT_REPO_ROOT/$OUT/two_classes.err
cd $GRAFT_REPO_ROOT
line $OUT/two_classes.json
f=$(find /tmp/prof2 -name "*kernel_stats.csv" | head -1); head -6 $f; cp $f $OUT/bench_kernel_stats_two_classes.csv
echo "== 3 classes under rocprofv3"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -o three -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-ginkgo-api --no-pmc --cg-iters 0 > $GRAFT_REPO_ROOT/$OUT/three_classes.json 2> $GRAFT_REPO_ROOT/$OUT/three_classes.err
cd $GRAFT_REPO_ROOT
line $OUT/three_classes.json
f=$(find /tmp/prof3 -name "*kernel_stats.csv" | head -1); head -4 $f; cp $f $OUT/bench_kernel_stats.csv
echo done
