#!/bin/bash
# round 2, session 11: matrix-core block-Jacobi (tests + timing), COO pass 1 with staged pointer bursts
TAG=${1:-r02s11}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_jacobi_mfma_gpu.py tests/test_coo_hybrid_gpu.py -q -x 2>&1 | tail -15
for t in matrix_coo_kernels_hip matrix_hybrid_kernels_hip; do
  timeout 300 oracle/_ref/dropin/reftests/$t > $OUT/$t.log 2>&1; echo "$t rc=$?"
  grep -E "^\[  PASSED  \]|tests ran|FAILED  \] [0-9]" $OUT/$t.log | head -4
done
timeout 600 python tools/jacobi_mfma_bench.py 256 2>&1 | grep -v amdgpu.ids | tee $OUT/jacobi_mfma_256.txt
timeout 600 python tools/format_bench.py 256 2>&1 | grep -v amdgpu.ids | tee $OUT/format_bench_256.txt
