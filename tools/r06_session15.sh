#!/bin/bash
# round 6, session 15: the two MPI reference suites that failed in the whole-suite run, alone, with their output;
# then with the round's anticipations off (GKOC_TUNE_10=2) to tell the binding from the MPI layer
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06s15
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/oracle/_ref/mpi_ga/reftests
for mode in default anticipate2; do
for n in distributed_matrix solver_solver; do
  np=3
  if [ $mode = anticipate2 ]; then export GKOC_TUNE_10=2; else unset GKOC_TUNE_10; fi
  s=$(date +%s)
  GKOC_MPI_TRANSPORT=rccl GKOC_TEST_RANK_LOG=$OUT/${n}_$mode timeout 400 /opt/conda/bin/mpiexec -n $np ./${n}_mpi_hip > $OUT/${n}_$mode.log 2>&1; rc=$?
  e=$(date +%s)
  ran=$(grep -o "^\[==========\] [0-9]* tests ran" $OUT/${n}_$mode.log | grep -o "[0-9]*" | head -1)
  fail=$(grep -o "^\[  FAILED  \] [0-9]* tests" $OUT/${n}_$mode.log | grep -o "[0-9]*" | head -1)
  echo "$mode $n rc=$rc ran=${ran:-?} failed=${fail:-0} in $((e-s)) s"
  grep -n "FAILED\|Fatal error\|RUN " $OUT/${n}_$mode.log | tail -6 | cut -c1-200
done
done
