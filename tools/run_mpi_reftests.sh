#!/bin/bash
# Ginkgo's own test/mpi binaries (oracle/build_mpi_dropin.py) under mpiexec with per-rank logs: gpurun_out/<tag>/; input of tools/update_mpi_reftests_expected.py
TAG=${1:-mpi_reftests}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/oracle/_ref/mpi_ga/reftests
for n in distributed_assembly distributed_matrix distributed_partition_helpers distributed_row_gatherer distributed_vector solver_solver preconditioner_schwarz; do
  np=3; [ $n = distributed_row_gatherer ] && np=6
  GKOC_TEST_RANK_LOG=$OUT/$n timeout 600 /opt/conda/bin/mpiexec -n $np ./${n}_mpi_hip > $OUT/${n}_mpi_hip.log 2>&1; rc=$?
  ran=$(grep -o "^\[==========\] [0-9]* tests ran" $OUT/${n}_mpi_hip.log | grep -o "[0-9]*" | head -1)
  pass=$(grep -o "^\[  PASSED  \] [0-9]* tests" $OUT/${n}_mpi_hip.log | grep -o "[0-9]*" | head -1)
  fail=$(grep -o "^\[  FAILED  \] [0-9]* tests" $OUT/${n}_mpi_hip.log | grep -o "[0-9]*" | head -1)
  echo "$n np=$np rc=$rc ran=${ran:-?} passed=${pass:-?} failed=${fail:-0}"
done | tee $OUT/summary.txt
exit 0
