#!/bin/bash
# round 6, session 24: jacobi::generate re-homes the block array (binding): parity, then the drop-in's CG with and without
OUT=gpurun_out/r06s24
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_dropin_gpu.py tests/test_reftests_gpu.py -m gpu -q 2>&1 | tail -5 | tee $OUT/parity.txt
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
for v in 1 0 1 0; do (cd $D && GKOC_TUNE_17=$v timeout 600 ./dropin_bench 256 50 200 --json 2>&1 | grep "solver::Cg" | sed "s/^/[rehome=$v] /"); done | tee $OUT/api_cg.txt
