#!/bin/bash
# round 2, session 9: unit-stride gather specialisation (CSR / ELL / SELL-P) on top of the new row loop
TAG=${1:-r02s9}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_spmv_gpu.py tests/test_flan_like_gpu.py tests/test_coo_hybrid_gpu.py -q -x 2>&1 | tail -3
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
timeout 600 python tools/format_bench.py 256 2>&1 | grep -v amdgpu.ids | tee $OUT/format_bench_256.txt
