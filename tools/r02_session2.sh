#!/bin/bash
# round-2 GPU session 2: arena through the Python mirror (tests + bench) and through Ginkgo's API
TAG=${1:-r02k}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_spmv_gpu.py -x -q 2>&1 | tail -8 | tee $OUT/pytest_spmv.txt
GKOC_ARENA_VERBOSE=1 timeout 600 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 3000 $OUT/bench.json
grep -c granule $OUT/bench.err
GKOC_ARENA=0 timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu > $OUT/bench_arena0.json 2> $OUT/bench_arena0.err; echo "bench arena0 rc=$?"; tail -c 1500 $OUT/bench_arena0.json
timeout 300 oracle/_ref/dropin/dropin_bench 256 50 100 > $OUT/dropin_bench.txt 2>&1; echo "dropin rc=$?"; tail -12 $OUT/dropin_bench.txt
GKOC_ARENA=0 timeout 300 oracle/_ref/dropin/dropin_bench 256 50 100 > $OUT/dropin_bench_arena0.txt 2>&1; echo "dropin arena0 rc=$?"; tail -12 $OUT/dropin_bench_arena0.txt
