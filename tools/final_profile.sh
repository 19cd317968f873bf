#!/bin/bash
# closing evidence of a round: smoke(), the dropin tests, the default bench line, and the same command under
# rocprofv3 --kernel-trace --stats (copy gpurun_out/<tag>/* to profiles/ afterwards)
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== dropin"
timeout 900 python -m pytest tests/test_dropin_gpu.py -m gpu -q 2>&1 | tail -3
(cd oracle/_ref/dropin && LD_LIBRARY_PATH=.:../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib timeout 600 ./dropin_test 2>&1 | grep -i "convert_to\|lookup\|by-product\|user precond\|FAIL" | head -8) | tee $OUT/dropin_lines.txt
echo "== driver's command"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_unprofiled.json 2> $OUT/bench.err
tail -1 $OUT/bench_line_unprofiled.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['pct_hbm_peak'], d.get('cg_iters_per_s'), d['roofline']['frac'], d['roofline']['traffic'], d['ginkgo_api']['frac'], d['ginkgo_api']['cg_iters_per_s'], d['cpu_baseline']['value'], d['placement']['memory_classes_found'], d['placement']['search_ms'])"
echo "== the same under rocprofv3"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -o b -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu --no-ginkgo-api > $GRAFT_REPO_ROOT/$OUT/bench_line_profiled.json 2> $GRAFT_REPO_ROOT/$OUT/bench_profiled.err
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_final -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-230; cp $f $OUT/bench_kernel_stats.csv
tail -1 $OUT/bench_line_profiled.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['pct_hbm_peak'], d.get('cg_iters_per_s'), d['roofline']['kernel_ms'])"
echo done
