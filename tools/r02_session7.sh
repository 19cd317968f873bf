#!/bin/bash
# round 2, session 7: lane-layout variants (E=1 / E=2) on Flan-like and L256, native driver host cost, rccl mirror test
TAG=${1:-r02s7}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 8 9; do
  GKOC_TUNE_2=$v timeout 600 python -m pytest tests/test_spmv_gpu.py -q -x 2>&1 | tail -1 | sed "s/^/layout$v: /"
done
for v in 0 8 9 0 8; do
  echo "== flan variant $v"
  GKOC_TUNE_2=$v timeout 600 python tools/flan_bench.py 80 2>&1 | grep -E "CSR SpMV" | tee -a $OUT/flan_layout$v.txt
done
for v in 0 8 9 0 8; do
  GKOC_TUNE_2=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu --cg-iters 0 > $OUT/bench_layout$v.json 2> $OUT/bench_layout$v.err
  python - <<PY
import json
l=[x for x in open("$OUT/bench_layout$v.json") if x.startswith("{")]
d=json.loads(l[-1]); print("L256 layout$v", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done
timeout 600 python -m pytest tests/test_distributed.py -q -x -m gpu -k rccl 2>&1 | tail -3
for sv in cg pipe_cg; do
  examples/native_dist_cg 16 3000 1e-30 $sv 8 mirror | grep "^{" | tee -a $OUT/native_host_cost.txt
done
examples/native_dist_cg 256 100 1e-30 cg 4 | grep "^{" | tee -a $OUT/native_l256.txt
