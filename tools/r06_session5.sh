#!/bin/bash
# round 6, session 5: bench line with the criterion -> step_1 anticipation, kernel traces of CbGmres (keep / complex)
# to see where 10-19 ms per iteration go, the 474-iteration solve against OmpExecutor
OUT=gpurun_out/r06s5
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench (default command)"
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "rc=$?"
tail -1 $OUT/bench_line.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('value',d['value'],'frac',r['frac'],'cg',d.get('cg_iters_per_s'),'gmres',d.get('gmres_iters_per_s'),'api',d.get('ginkgo_api',{}).get('cg_iters_per_s'), d.get('ginkgo_api',{}).get('one_kernel_per_call'))
"
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
for W in cbd-keep cbd-reduce1 cbc; do
  echo "== trace $W"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/tr_$W -o t -- $D/round5_bench 256 30 $W > $GRAFT_REPO_ROOT/$OUT/tr_$W.log 2>&1)
  tail -3 $OUT/tr_$W.log
  f=$(find $OUT/tr_$W -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -25 "$f" | cut -c1-220 | tee $OUT/tr_${W}_kernel_stats.txt
  find $OUT/tr_$W -name '*.db' -delete; find $OUT/tr_$W -name '*kernel_trace.csv' -size +20M -delete
done
echo "== full solve vs OmpExecutor"
GKO_TEST_FULL_SOLVE=1 timeout 1500 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "reference_omp_executor" -s 2>&1 | tail -15 | tee $OUT/full_solve.txt
