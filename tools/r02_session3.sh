#!/bin/bash
TAG=${1:-r02v}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err); echo "prof rc=$?"
find $OUT/prof -name "*kernel_stats*" | head -1 | xargs -r head -14 | cut -c1-200
GKO_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err; echo "forcedist rc=$?"; tail -c 1200 $OUT/bench_forcedist.json; tail -3 $OUT/bench_forcedist.err
timeout 1400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
