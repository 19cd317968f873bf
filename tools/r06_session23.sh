#!/bin/bash
# round 6, closing evidence: tools/final_profile.sh, the other workloads of the bench, format / dtype / multi-column tables
OUT=gpurun_out/r06final
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/final_profile.sh r06final 2>&1 | tee $OUT/final_profile.log
echo "== other workloads"
timeout 600 python bench.py --workload flan --no-cpu --no-pmc --no-ginkgo-api 2>/dev/null | tail -1 > $OUT/bench_flan.json; python -c "
import json; d=json.loads(open('$OUT/bench_flan.json').read()); print('flan', d['value'], d['unit'], d['roofline']['frac'], d.get('cg_iters_per_s'))"
timeout 600 python bench.py --workload irregular --no-cpu --no-pmc --no-ginkgo-api 2>/dev/null | tail -1 > $OUT/bench_irregular.json; python -c "
import json; d=json.loads(open('$OUT/bench_irregular.json').read()); print('irregular', d['value'], d['unit'], d['roofline']['frac'], d.get('cg_iters_per_s'))"
echo "== tables"
timeout 600 python tools/format_bench.py 256 2>&1 | tail -8 | tee $OUT/formats.txt
timeout 600 python tools/flan_bench.py 80 2>&1 | tail -5 | tee $OUT/flan.txt
timeout 600 python tools/dtype_bench.py 2>&1 | tail -16 | tee $OUT/dtype.txt
FORMATS=csr timeout 600 python tools/multi_rhs_bench.py 256 2>&1 | tail -6 | tee $OUT/multi_rhs.txt
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
(cd $D && timeout 900 ./round5_bench 256 30 jacobi 2>&1 | tail -7) | tee $OUT/round5_additions_jacobi.txt
for W in cbd-keep cbd-reduce1 cbc; do (cd $D && timeout 600 ./round5_bench 256 30 $W 2>&1 | grep CbGmres); done | tee $OUT/round5_additions_cb.txt
