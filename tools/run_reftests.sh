#!/bin/bash
# run Ginkgo's own tests (oracle/_ref/dropin/reftests/*_hip) on this backend; summary per suite
OUT=${1:-gpurun_out/reftests}
mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-.}
for t in oracle/_ref/dropin/reftests/*_hip; do
  n=$(basename $t)
  timeout 300 $t > $OUT/$n.log 2>&1
  rc=$?
  ran=$(grep -o "^\[==========\] [0-9]* tests ran" $OUT/$n.log | grep -o "[0-9]*")
  pass=$(grep -o "^\[  PASSED  \] [0-9]* tests" $OUT/$n.log | grep -o "[0-9]*")
  skip=$(grep -o "^\[  SKIPPED \] [0-9]* tests" $OUT/$n.log | grep -o "[0-9]*")
  fail=$(grep -o "^\[  FAILED  \] [0-9]* tests" $OUT/$n.log | grep -o "[0-9]*")
  echo "$n rc=$rc ran=${ran:-?} passed=${pass:-?} skipped=${skip:-0} failed=${fail:-0}"
done | tee $OUT/summary.txt
