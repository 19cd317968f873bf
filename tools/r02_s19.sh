cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_coo_hybrid_gpu.py tests/test_spmv_gpu.py -q -x 2>&1 | tail -6
for t in matrix_coo_kernels_hip matrix_hybrid_kernels_hip; do
  timeout 300 oracle/_ref/dropin/reftests/$t 2>&1 | grep -E "^\[  PASSED  \]|tests ran|FAILED  \] [0-9]" | head -3
done
timeout 600 python tools/format_bench.py 256 2>&1 | grep -E " csr | coo |hybrid"
GKOC_TUNE_4=0 timeout 600 python tools/format_bench.py 256 2>&1 | grep -E " coo " | sed "s/^/two-pass: /"
