#!/bin/bash
TAG=${1:-r03s12}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== default bench command (with live counters)"
SECONDS=0; timeout 900 python bench.py 2> $OUT/bench_default.err | grep '^{"metric"' | tail -1 > $OUT/bench_line.json
echo "bench.py took $SECONDS s"
python - <<PY
import json
d=json.load(open("$OUT/bench_line.json"))
print({k:d[k] for k in ("value","ms_per_step","cg_iters_per_s")})
print(json.dumps(d["roofline"], indent=1))
PY
grep -v "amdgpu.ids" $OUT/bench_default.err | tail -5
exit 0
