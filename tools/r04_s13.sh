#!/bin/bash
# round 4, session 13: complex Krylov / ELL / SELL-P kernels behind Ginkgo's own suites
TAG=${1:-r04s13}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== reference suites"
timeout 2400 python -m pytest tests/test_reftests_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -60 | tee $OUT/reftests.txt
echo "== MPI reference suites"
timeout 2400 python -m pytest tests/test_mpi_reftests_gpu.py tests/test_mpi_dropin_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -40 | tee $OUT/mpi_reftests.txt
echo "== dropin + krylov family"
timeout 1500 python -m pytest tests/test_dropin_gpu.py tests/test_krylov_family_gpu.py tests/test_gmres_gpu.py -m gpu -q 2>&1 | tail -5 | tee $OUT/others.txt
echo done
