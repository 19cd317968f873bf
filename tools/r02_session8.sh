#!/bin/bash
# round 2, session 8: new row-phase loop (Flan-like + L256 + bit-exactness), native driver host-time breakdown
TAG=${1:-r02s8}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_spmv_gpu.py tests/test_flan_like_gpu.py tests/test_native_cg_gpu.py -q -x 2>&1 | tail -4
timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -x -k "oracle_at_full_size or formats_agree" 2>&1 | tail -3
for i in 1 2; do
  timeout 600 python tools/flan_bench.py 80 2>&1 | grep -E "SpMV|it/s" | tee -a $OUT/flan.txt
done
for i in 1 2; do
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  python - <<PY
import json
l=[x for x in open("$OUT/bench_$i.json") if x.startswith("{")]
d=json.loads(l[-1]); print("L256", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("cg_iters_per_s"))
PY
done
for sv in cg pipe_cg; do
  GKOC_EXAMPLE_TRACE=1 examples/native_dist_cg 16 3000 1e-30 $sv 8 mirror 2>&1 | grep -E "^\{|host us" | tee -a $OUT/native_host_cost.txt
done
GKOC_EXAMPLE_TRACE=1 examples/native_dist_cg 16 3000 1e-30 cg 8 2>&1 | grep -E "^\{|host us" | tee -a $OUT/native_host_cost.txt
timeout 300 python tools/dist_host_cost.py 16 400 direct 2>&1 | grep -E "grid|all_reduce|exchange" | tee $OUT/python_host_cost.txt
