#!/bin/bash
OUT=gpurun_out/r06s27
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_dropin_gpu.py -m gpu -q 2>&1 | tail -4 | tee $OUT/parity.txt
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
(cd $D && timeout 900 ./dropin_test 2>&1 | grep -i "FAIL\|Jacobi blocks\|re-hom\|DROPIN" | head -8) | tee $OUT/dropin_lines.txt
