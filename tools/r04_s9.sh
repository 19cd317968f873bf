#!/bin/bash
# round 4, session 9: by-products behind the unmodified Ginkgo core (dropin tests, Ginkgo's own suites), the
# default bench line with its Ginkgo-API leg
TAG=${1:-r04s9}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== dropin + reference suites"
timeout 1500 python -m pytest tests/test_dropin_gpu.py tests/test_reftests_gpu.py -m gpu -q -x 2>&1 | tail -15 | tee $OUT/tests.txt
(cd oracle/_ref/dropin && LD_LIBRARY_PATH=.:../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib timeout 600 ./dropin_test 2>&1 | grep -i "by-product\|user precond\|FAIL\|passed\|failed" | head -20) | tee $OUT/dropin_byproducts.txt
echo "== Ginkgo-API CG, three modes, 2 runs each"
for m in 0 2 1 0 2 1; do
(cd oracle/_ref/dropin && GKOC_TUNE_5=$m LD_LIBRARY_PATH=.:../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib timeout 600 ./dropin_bench 256 20 200 --json 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mode $m', d['csr_apply_ms'], d['cg_iters_per_s'], d['cg_iterations'])")
done | tee $OUT/ginkgo_api_modes.txt
echo "== default bench line"
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['pct_hbm_peak'], d.get('cg_iters_per_s'), d.get('ginkgo_api'), d['roofline'], d['cpu_baseline'])"
echo done
