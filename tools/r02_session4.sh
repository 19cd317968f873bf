#!/bin/bash
# round 2, session 4: cb_gmres through Ginkgo's own test + solver, CSR ring variants (bit-exactness, Flan-like, L256)
TAG=${1:-r02s4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for t in solver_cb_gmres_kernels_hip solver_idr_kernels_hip solver_gmres_kernels_hip; do
  timeout 300 oracle/_ref/dropin/reftests/$t > $OUT/$t.log 2>&1; echo "$t rc=$?"
  grep -E "^\[  (PASSED|FAILED)  \]|tests ran" $OUT/$t.log | head -12
done
(cd oracle/_ref/dropin && timeout 600 ./dropin_test 24 > $OUT/dropin_test.log 2>&1); echo "dropin_test rc=$?"
grep -E "CbGmres|FAILED|DROPIN" $OUT/dropin_test.log | head -40
for v in 1 2 3 4; do
  GKOC_TUNE_2=$v timeout 600 python -m pytest tests/test_spmv_gpu.py -q -x 2>&1 | tail -2 | sed "s/^/ring$v: /"
done
for v in 0 1 2 3 4; do
  echo "== flan ring variant $v"
  GKOC_TUNE_2=$v timeout 600 python tools/flan_bench.py 80 2>&1 | grep -E "CSR SpMV|SELL-P SpMV|it/s|iterations" | tee -a $OUT/flan_ring$v.txt
done
for v in 0 1 4; do
  echo "== L256 ring variant $v"
  GKOC_TUNE_2=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu > $OUT/bench_ring$v.json 2> $OUT/bench_ring$v.err
  python - <<PY
import json
l=[x for x in open("$OUT/bench_ring$v.json") if x.startswith("{")]
d=json.loads(l[-1]); print("ring$v", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("cg"))
PY
done
