#!/bin/bash
# Ginkgo's distributed classes on this backend under mpiexec (all ranks on GPU 0)
OUT=${1:-gpurun_out/mpi}
mkdir -p $OUT
OUT=$(realpath $OUT)
cd ${GRAFT_REPO_ROOT:-.}/oracle/_ref/mpi/bin
MPIEXEC=${MPIEXEC:-/opt/conda/bin/mpiexec}
timeout 300 $MPIEXEC -n 2 ./mpi_dist_test 24 > $OUT/mpi_dist_test_n2.txt 2>&1; echo "mpi_dist_test n2 rc=$?"
timeout 300 $MPIEXEC -n 3 ./mpi_dist_test 20 > $OUT/mpi_dist_test_n3.txt 2>&1; echo "mpi_dist_test n3 rc=$?"
timeout 300 $MPIEXEC -n 2 ./distributed-solver hip 2000 > $OUT/distributed_solver_hip.txt 2>&1; echo "distributed-solver hip rc=$?"
timeout 300 $MPIEXEC -n 2 ./distributed-solver reference 2000 > $OUT/distributed_solver_ref.txt 2>&1; echo "distributed-solver reference rc=$?"
tail -n 22 $OUT/*.txt
