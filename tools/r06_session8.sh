#!/bin/bash
# round 6, session 8: parity of what sessions 6-7 touched (the parity step of session 7 named a file that does not
# exist and ran nothing), complex CSR SpMV through the row-segment kernel (+ CB-GMRES complex again), the timeline of
# the unmodified core's CG on the drop-in
OUT=gpurun_out/r06s8
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity"
timeout 2400 python -m pytest tests/test_spmv_gpu.py tests/test_flan_like_gpu.py tests/test_jacobi_types_gpu.py tests/test_krylov_gpu.py tests/test_complex_gpu.py tests/test_mpi_dropin_gpu.py tests/test_dropin_gpu.py -m gpu -q 2>&1 | tail -15 | tee $OUT/parity.txt
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
echo "== complex CB-GMRES (new complex CSR SpMV), then with GKOC_TUNE_16=1 (thread per row)"
(cd $D && timeout 600 ./round5_bench 256 30 cbc 2>&1 | tail -2) | tee $OUT/cbc_new.txt
(cd $D && GKOC_TUNE_16=1 timeout 600 ./round5_bench 256 30 cbc 2>&1 | tail -2) | tee $OUT/cbc_old.txt
echo "== api gap"
bash tools/api_gap.sh r06s8/api_gap > /dev/null 2>&1
sed -n 1,60p $OUT/api_gap/report.txt
tail -3 $OUT/api_gap/plain.txt | cut -c1-600
rm -rf $OUT/api_gap/trace
