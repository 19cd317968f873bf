#!/bin/bash
# round 6, session 19: kernel stats of CbGmres<double> (keep, reduce1) at 256^3 - where do 3.0 / 2.3 ms per iteration go?
OUT=gpurun_out/r06s19
mkdir -p $OUT
export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
for W in cbd-keep cbd-reduce1; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$W -o c -- $D/round5_bench 256 30 $W > $GRAFT_REPO_ROOT/$OUT/$W.log 2>&1)
  cp $(find /tmp/prof_$W -name '*kernel_stats.csv' | head -1) $OUT/${W}_kernel_stats.csv
  tail -1 $OUT/$W.log
done
