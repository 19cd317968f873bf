#!/bin/bash
OUT=gpurun_out/r06s10
mkdir -p $OUT
export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
echo "== api gap with the host's calls"
bash tools/api_gap.sh r06s10/api_gap > /dev/null 2>&1
grep -n "two iterations, host and device" -A90 $OUT/api_gap/report.txt
rm -rf $OUT/api_gap/trace
echo "== complex CB-GMRES with the kernel bounded to four waves per SIMD"
(cd $D && timeout 600 ./round5_bench 256 30 cbc 2>&1 | tail -2) | tee $OUT/cbc_new.txt
