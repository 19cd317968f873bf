#!/bin/bash
TAG=${1:-r03s5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== MPI tests (both flavors)"
timeout 1500 python -m pytest tests/test_mpi_dropin_gpu.py -m gpu -q -x 2>&1 | tail -30 | tee $OUT/pytest_mpi.txt
echo "== ga flavor output n=2"
(cd oracle/_ref/mpi_ga/bin && GKOC_MPI_VERBOSE=1 timeout 300 /opt/conda/bin/mpiexec -n 2 ./mpi_dist_test 24 2>&1 | tail -40) | tee $OUT/mpi_ga_n2.txt
echo "== ga flavor n=1 rccl"
(cd oracle/_ref/mpi_ga/bin && GKOC_MPI_VERBOSE=1 GKOC_MPI_MODE=rccl timeout 300 /opt/conda/bin/mpiexec -n 1 ./mpi_dist_test 20 2>&1 | tail -12) | tee $OUT/mpi_ga_n1_rccl.txt
exit 0
