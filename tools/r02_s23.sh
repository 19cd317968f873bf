#!/bin/bash
OUT=gpurun_out/r02s23; mkdir -p $OUT
cd oracle/_ref/dropin/reftests
for s in preconditioner_jacobi_kernels_hip matrix_dense_kernels_hip matrix_csr_kernels2_hip matrix_ell_kernels_hip matrix_sellp_kernels_hip matrix_hybrid_kernels_hip matrix_coo_kernels_hip components_fill_array_kernels_hip base_device_matrix_data_kernels_hip solver_idr_kernels_hip solver_bicg_kernels_hip solver_minres_kernels_hip; do
  timeout 300 ./$s > ../../../../$OUT/$s.txt 2>&1
  echo "$s rc=$?"; grep -c "^\[  FAILED  \]" ../../../../$OUT/$s.txt
done
exit 0
