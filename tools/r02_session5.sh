#!/bin/bash
# round 2, session 5: COO one-pass row pointers (tests + timing), Flan-like CSR with 32-row / U=2 variants
TAG=${1:-r02s5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_coo_hybrid_gpu.py -q -x 2>&1 | tail -5
for t in matrix_coo_kernels_hip matrix_hybrid_kernels_hip; do
  timeout 300 oracle/_ref/dropin/reftests/$t > $OUT/$t.log 2>&1; echo "$t rc=$?"
  grep -E "^\[  PASSED  \]|tests ran|FAILED  \] [0-9]" $OUT/$t.log | head -4
done
timeout 600 python tools/format_bench.py 256 2>&1 | tee $OUT/format_bench_256.txt
for v in 5 6 7 0; do
  echo "== flan ring variant $v"
  GKOC_TUNE_2=$v timeout 600 python tools/flan_bench.py 80 2>&1 | grep -E "CSR SpMV|SELL-P SpMV" | tee -a $OUT/flan_ring$v.txt
done
for v in 5 6 7; do
  GKOC_TUNE_2=$v timeout 600 python -m pytest tests/test_spmv_gpu.py -q -x 2>&1 | tail -1 | sed "s/^/ring$v: /"
done
