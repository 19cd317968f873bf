#!/bin/bash
# round 6, session 16: the two MPI reference suites again (complex product falls back on null arrays; spmv + dot
# needs two confirmations and forgets freed operands); level 3 = everything but the product + dot
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06s16
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/oracle/_ref/mpi_ga/reftests
for mode in default level3; do
for n in solver_solver distributed_matrix; do
  [ $mode = level3 ] && [ $n = distributed_matrix ] && continue
  if [ $mode = level3 ]; then export GKOC_TUNE_10=3; else unset GKOC_TUNE_10; fi
  s=$(date +%s)
  GKOC_MPI_TRANSPORT=rccl GKOC_TEST_RANK_LOG=$OUT/${n}_$mode timeout 300 /opt/conda/bin/mpiexec -n 3 ./${n}_mpi_hip > $OUT/${n}_$mode.log 2>&1; rc=$?
  e=$(date +%s)
  ran=$(grep -o "^\[==========\] [0-9]* tests ran" $OUT/${n}_$mode.log | grep -o "[0-9]*" | head -1)
  fail=$(grep -o "^\[  FAILED  \] [0-9]* tests" $OUT/${n}_$mode.log | grep -o "[0-9]*" | head -1)
  echo "$mode $n rc=$rc ran=${ran:-?} failed=${fail:-0} in $((e-s)) s"
  grep -n "FAILED\|Fatal error\|RUN " $OUT/${n}_$mode.log | tail -5 | cut -c1-200
done
done
