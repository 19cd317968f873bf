#!/bin/bash
# round 6, session 17: all seven MPI reference suites (default settings), then the drop-in test program
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06s17
mkdir -p $OUT
export TMPDIR=/tmp
export GKOC_MPI_TRANSPORT=rccl
bash tools/run_mpi_reftests.sh r06s17/mpi
cd $GRAFT_REPO_ROOT
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
(cd $D && timeout 900 ./dropin_test 2>&1 | grep -i "FAIL\|passed\|failed\|anticipated:\|by-products:" | head -12) | tee $OUT/dropin_lines.txt
timeout 1200 python -m pytest tests/test_mpi_reftests_gpu.py tests/test_mpi_dropin_gpu.py tests/test_dropin_gpu.py tests/test_native_cg_gpu.py -m gpu -q 2>&1 | tail -6 | tee $OUT/parity.txt
