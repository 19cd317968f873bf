#!/bin/bash
# round 6, session 20: CB-GMRES passes with eight (complex: four) basis vectors under way: parity, then timings
OUT=gpurun_out/r06s20
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_reftests_gpu.py tests/test_dropin_gpu.py tests/test_krylov_family_gpu.py -m gpu -q 2>&1 | tail -6 | tee $OUT/parity.txt
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
for W in cbd-keep cbd-reduce1 cbc; do (cd $D && timeout 600 ./round5_bench 256 30 $W 2>&1 | tail -2 | grep CbGmres); done | tee $OUT/cb_gmres.txt
