#!/bin/bash
# round 4, session 2: the galloping survey + forced 2 / 1 class outcomes
TAG=${1:-r04s2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== arena classes test"
timeout 900 python -m pytest tests/test_arena_classes_gpu.py -x -q -s 2>&1 | tail -25 | tee $OUT/arena_classes.txt
echo "== quick bench x3, verbose arena"
for i in 1 2 3; do
GKOC_ARENA_VERBOSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-ginkgo-api --no-pmc --cg-iters 30 > $OUT/bench_$i.json 2> $OUT/bench_$i.err
tail -1 $OUT/bench_$i.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['placement']; print(d['value'], d['config']['pct_hbm_peak'], d.get('cg_iters_per_s'), p['memory_classes_found'], p['class_of'], p['granules_walked'], p['granules_classified'], p['search_ms'], p['probe_retries'])"
grep "granule\|survey" $OUT/bench_$i.err | awk '/granule/ {printf "%s", $6} /survey/ {print ""; print}'
done
echo "== spmv + arena role tests"
timeout 900 python -m pytest tests/test_spmv_gpu.py tests/test_arena_roles_gpu.py tests/test_dropin_gpu.py -x -q 2>&1 | tail -5
echo done
