#!/bin/bash
# round 6, session 21: is CbGmres' 2.9 TB/s in the Arnoldi passes an aliasing effect of 2^27-byte vectors (256^3 x 8 B)?
# the same solver on 250^3 and 252^3; and the Gmres tests with the batched multi-dot
OUT=gpurun_out/r06s21
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gmres_gpu.py tests/test_complex_gpu.py -m gpu -q 2>&1 | tail -4 | tee $OUT/parity.txt
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
for G in 256 252 250 200; do (cd $D && timeout 600 ./round5_bench $G 30 cbd-keep 2>&1 | grep "CbGmres\|27-pt"); done | tee $OUT/cb_gmres_sizes.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_250 -o c -- $D/round5_bench 250 30 cbd-keep > /dev/null 2>&1)
f=$(find /tmp/prof_250 -name '*kernel_stats.csv' | head -1); grep "cb_dots_stage1\|cb_update_stage1\|csr_spmv" $f | cut -c1-60,300-420 | tee $OUT/stats_250.txt
