#!/bin/bash
# round 3, session 1: the new world-8 tests, full-size parity vs the reference's OmpExecutor, the
# bench line with ginkgo_api + the bound OpenMP baseline, per-rank device cost of the 8-rank run
# (with a kernel trace)
TAG=${1:-r03s1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== box"; nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2 | tail -1
echo "== distributed tests (incl. 8 ranks on one GPU)"
timeout 1500 python -m pytest tests/test_distributed.py -m gpu -q -x --durations=8 2>&1 | tail -25 | tee $OUT/pytest_distributed.txt
echo "== natural failure: RcclComm with 2 ranks on one device (no injection)"
GKO_COMM=rccl timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29655 tests/dist_worker.py fallback-gpu 0 > $OUT/natural_fallback.txt 2>&1; echo "rc=$?"; grep -v Gloo $OUT/natural_fallback.txt | tail -6
echo "== dropin_test (fusion opt-in, small user blocks)"
timeout 600 python -m pytest tests/test_dropin_gpu.py -m gpu -q -x 2>&1 | tail -5
(cd oracle/_ref/dropin && timeout 300 ./dropin_test 24 2>&1 | grep -i "fail\|small user\|fusion" | head -20)
echo "== full-size parity vs reference omp"
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -s -k "reference_omp" 2>&1 | tail -15 | tee $OUT/fullsize_ref.txt
echo "== default bench"
timeout 900 python bench.py 2> $OUT/bench_default.err | grep '^{"metric"' | tail -1 > $OUT/bench_line.json; python - <<PY
import json
d=json.load(open("$OUT/bench_line.json"))
print({k:d[k] for k in ("value","ms_per_step","cg_iters_per_s")}, d["roofline"]["frac"])
print("ginkgo_api:", d.get("ginkgo_api"))
print("cpu_baseline:", d.get("cpu_baseline"))
PY
tail -3 $OUT/bench_default.err
echo "== dist_sim 256 8 3"
timeout 600 python tools/dist_sim.py 256 8 3 200 2>&1 | tee $OUT/dist_sim_256_8.txt | tail -14
echo "== dist_sim under kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ds -- python $GRAFT_REPO_ROOT/tools/dist_sim.py 256 8 3 100 > $OUT/trace_run.txt 2>&1)
find $OUT/trace -name "*kernel_stats*" | head -1 | xargs -r -I{} cp {} $OUT/dist_sim_kernel_stats.csv
rm -rf $OUT/trace
python - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/dist_sim_kernel_stats.csv")))[:28]:
    print(r['Calls'], f"{float(r['AverageNs'])/1e3:9.1f} us  tot {float(r['TotalDurationNs'])/1e6:8.2f} ms", r['Name'][:110])
PY
exit 0
