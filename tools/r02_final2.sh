#!/bin/bash
# round 2, closing run after the step_2 + block-Jacobi fusion: full GPU suite, default bench line,
# the same under rocprofv3 --kernel-trace --stats, per-rank device cost of the 8-rank runs
TAG=${1:-r02final2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
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
echo "== per-rank device cost, 256^3 over 8 ranks (strong scaling) and 512^3 over 8 ranks (config 3)"
timeout 600 python tools/dist_sim.py 256 8 3 200 2>&1 | grep -v amdgpu.ids | tee $OUT/dist_sim_256_8.txt
timeout 600 python tools/dist_sim.py 512 8 3 100 2>&1 | grep -v amdgpu.ids | tee $OUT/dist_sim_512_8.txt
echo "== Ginkgo API"
(cd oracle/_ref/dropin && timeout 600 ./dropin_bench 256 50 100 2>&1 | tee $OUT/ginkgo_api_bench.txt | tail -5)
