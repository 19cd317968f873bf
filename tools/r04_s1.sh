#!/bin/bash
# round 4, session 1: how the driver hands out the memory classes; reproduce round 3's failed search
TAG=${1:-r04s1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== vram state"; cat /sys/class/drm/card*/device/mem_info_vram_used 2>/dev/null | head -8
echo "== fresh bench (old library)"
GKOC_ARENA_VERBOSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-ginkgo-api --no-pmc --cg-iters 0 > $OUT/bench_fresh.json 2> $OUT/bench_fresh.err
tail -1 $OUT/bench_fresh.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['pct_hbm_peak'], d['placement']['memory_classes_found'], d['placement']['class_of'], d['placement']['granules_walked'])"
grep -c granule $OUT/bench_fresh.err
echo "== survey, fresh"
timeout 300 tools/class_lab survey 24 2>&1 | tee $OUT/survey_fresh.txt
echo "== starve"
tools/class_lab starve 140 > $OUT/starve.txt 2>&1 &
HOLDER=$!
for i in $(seq 1 120); do grep -q READY $OUT/starve.txt && break; sleep 1; done
cat $OUT/starve.txt
echo "== bench under starvation (old library)"
GKOC_ARENA_VERBOSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-ginkgo-api --no-pmc --cg-iters 0 > $OUT/bench_starved.json 2> $OUT/bench_starved.err
tail -1 $OUT/bench_starved.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['pct_hbm_peak'], d['placement']['memory_classes_found'], d['placement']['class_of'], d['placement']['granules_walked'])"
grep "granule" $OUT/bench_starved.err | awk '{printf "%s", $6} END {print ""}'
echo "== survey under starvation"
timeout 300 tools/class_lab survey 8 2>&1 | tee $OUT/survey_starved.txt
kill $HOLDER
wait $HOLDER 2>/dev/null
echo done
