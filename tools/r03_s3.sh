#!/bin/bash
TAG=${1:-r03s3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== new reference suites"
for n in distributed_assembly_kernels distributed_index_map_kernels distributed_matrix_kernels distributed_partition_helper_kernels distributed_partition_kernels distributed_vector_kernels matrix_matrix solver_solver matrix_permutation_kernels matrix_scaled_permutation_kernels matrix_diagonal_kernels matrix_sparsity_csr_kernels components_precision_conversion_kernels components_reduce_array_kernels components_absolute_array_kernels; do
  t=oracle/_ref/dropin/reftests/${n}_hip
  (cd oracle/_ref/dropin/reftests && timeout 600 ./${n}_hip > $OUT/$n.log 2>&1); rc=$?
  ran=$(grep -o "^\[==========\] [0-9]* tests ran" $OUT/$n.log | grep -o "[0-9]*")
  pass=$(grep -o "^\[  PASSED  \] [0-9]* tests" $OUT/$n.log | grep -o "[0-9]*")
  fail=$(grep -o "^\[  FAILED  \] [0-9]* tests" $OUT/$n.log | grep -o "[0-9]*")
  echo "$n rc=$rc ran=${ran:-?} passed=${pass:-?} failed=${fail:-0}"
done | tee $OUT/summary.txt
echo "== 8 ranks 64"
timeout 900 python -m pytest tests/test_distributed.py -m gpu -q -x -k "eight_ranks" 2>&1 | tail -5
exit 0
