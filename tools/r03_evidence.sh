#!/bin/bash
# round 3: smoke, the default bench line (with its live counter passes), the same command under
# rocprofv3 --kernel-trace --stats
TAG=${1:-r03evidence}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== driver's bench command"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench_default.err | grep '^{"metric"' | tail -1 > $OUT/bench_line.json
python - <<PY
import json
d=json.load(open("$OUT/bench_line.json"))
print({k:d[k] for k in ("value","ms_per_step","cg_iters_per_s")}, d["roofline"])
print("ginkgo_api:", {k:v for k,v in d.get("ginkgo_api",{}).items() if k in ("csr_apply_ms","frac","cg_iters_per_s","with_fusion_across_calls")})
print("cpu_baseline:", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("host_triad_gbs"))
PY
echo "== the same under rocprofv3 --kernel-trace --stats"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu --no-ginkgo-api 2> $OUT/bench_profiled.err | grep '^{"metric"' | tail -1 > $OUT/bench_line_profiled.json
cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp $f $OUT/bench_kernel_stats.csv
head -8 $OUT/bench_kernel_stats.csv | cut -c1-200
python -c "
import json; d=json.load(open('$OUT/bench_line_profiled.json')); print('profiled run:', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
rm -rf $OUT/prof
exit 0
