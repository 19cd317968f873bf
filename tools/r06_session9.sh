#!/bin/bash
# round 6, session 9: pinned-flag visibility lab; waves per workgroup / XCD order A/B; complex CSR SpMV alone
OUT=gpurun_out/r06s9
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pinned flag lab"
timeout 120 tools/lab_bin/pinned_flag_lab 2>&1 | tee $OUT/pinned_flag_lab.txt
echo "== waves per workgroup / xcd order"
timeout 1200 python tools/wpb_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/wpb_ab.txt
echo "== complex csr spmv alone (kernel trace)"
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o c -- $D/round5_bench 256 30 cbc > $GRAFT_REPO_ROOT/$OUT/cbc.log 2>&1)
cp $(find /tmp/prof_c -name '*kernel_stats.csv' | head -1) $OUT/cbc_kernel_stats.csv
head -8 $OUT/cbc_kernel_stats.csv | cut -c1-120,300-420
