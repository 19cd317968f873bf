#!/bin/bash
# round 3: the whole GPU suite (as the driver runs it), smoke, default bench line
TAG=${1:-r03full}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q -x --durations=12 2>&1 | tail -40 | tee $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== default bench command"
timeout 900 python bench.py 2> $OUT/bench_default.err | grep '^{"metric"' | tail -1 > $OUT/bench_line.json
python - <<PY
import json
d=json.load(open("$OUT/bench_line.json"))
print({k:d[k] for k in ("value","ms_per_step","cg_iters_per_s")}, d["roofline"]["frac"], d.get("placement",{}).get("class_of"))
print("ginkgo_api:", {k:v for k,v in d.get("ginkgo_api",{}).items() if k in ("csr_apply_ms","frac","cg_iters_per_s","with_fusion_across_calls","memory_classes")})
print("cpu_baseline:", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("host_triad_gbs"))
PY
exit 0
