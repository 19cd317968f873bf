#!/bin/bash
TAG=${1:-r03s4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in matrix_matrix solver_solver matrix_diagonal_kernels matrix_sparsity_csr_kernels components_precision_conversion_kernels components_reduce_array_kernels components_absolute_array_kernels matrix_dense_kernels solver_bicg_kernels; do
  (cd oracle/_ref/dropin/reftests && timeout 600 ./${n}_hip > $OUT/$n.log 2>&1); rc=$?
  ran=$(grep -o "^\[==========\] [0-9]* tests ran" $OUT/$n.log | grep -o "[0-9]*")
  pass=$(grep -o "^\[  PASSED  \] [0-9]* tests" $OUT/$n.log | grep -o "[0-9]*")
  fail=$(grep -o "^\[  FAILED  \] [0-9]* tests" $OUT/$n.log | grep -o "[0-9]*")
  echo "$n rc=$rc ran=${ran:-?} passed=${pass:-?} failed=${fail:-0}"
done | tee $OUT/summary.txt
exit 0
