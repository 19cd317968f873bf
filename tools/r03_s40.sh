#!/bin/bash
OUT=gpurun_out/${1:-r03s40}/reftests
mkdir -p $OUT
for n in stop_residual_norm_kernels_hip matrix_csr_kernels2_hip matrix_diagonal_kernels_hip components_absolute_array_kernels_hip components_reduce_array_kernels_hip components_precision_conversion_kernels_hip; do
  timeout 300 oracle/_ref/dropin/reftests/$n > $OUT/$n.log 2>&1
  echo "$n rc=$? $(grep -o '^\[  PASSED  \] [0-9]* tests' $OUT/$n.log) $(grep -o '^\[  FAILED  \] [0-9]* tests' $OUT/$n.log)"
  grep "^\[  FAILED  \] [A-Z]" $OUT/$n.log | sort -u | head -12
done
