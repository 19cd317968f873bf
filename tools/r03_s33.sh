#!/bin/bash
# final multi-RHS table + counters of the final kernels + the GPU tests that touch them
TAG=${1:-r03s33}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests/test_spmv_gpu.py tests/test_coo_hybrid_gpu.py tests/test_gmres_gpu.py tests/test_krylov_gpu.py tests/test_reftests_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/multi_rhs_bench.py 256 > $OUT/multi_rhs_256.txt 2>&1
cat $OUT/multi_rhs_256.txt
bash tools/multi_pmc.sh $TAG/pmc > /dev/null 2>&1
grep -c "" $OUT/pmc/multi_pmc_summary.txt
