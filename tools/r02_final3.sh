#!/bin/bash
# round 2, closing run: full GPU suite, Ginkgo's own test binaries, default bench line, the same under
# rocprofv3 --kernel-trace --stats, Ginkgo API bench
TAG=${1:-r02final3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_gpu.txt
echo "== Ginkgo's own test binaries"
bash tools/run_reftests.sh $OUT/reftests > /dev/null 2>&1
grep -v "failed=0" $OUT/reftests/summary.txt
echo "== default bench command"
timeout 900 python bench.py 2> $OUT/bench_default.err | grep '^{"metric"' | tail -1 | tee $OUT/bench_line_unprofiled.json | cut -c1-300
echo "== under rocprofv3 --kernel-trace --stats"
(cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu > $OUT/trace_run.txt 2>&1)
grep '^{"metric"' $OUT/trace_run.txt | tail -1 > $OUT/bench_line_profiled.json
find $OUT/trace -name "*kernel_stats*" | head -1 | xargs -r -I{} cp {} $OUT/bench_kernel_stats.csv
rm -rf $OUT/trace
python - <<PY
import csv, json
for r in list(csv.DictReader(open("$OUT/bench_kernel_stats.csv")))[:7]:
    print(r['Calls'], f"{float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f}", r['Name'][:100])
d = json.loads(open("$OUT/bench_line_profiled.json").read())
print("profiled line:", d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("cg_iters_per_s"))
PY
echo "== Ginkgo API"
(cd oracle/_ref/dropin && timeout 600 ./dropin_bench 256 50 100 2>&1 | tee $OUT/ginkgo_api_bench.txt | tail -5)
exit 0
