#!/bin/bash
# complex BLAS-1: Ginkgo's test/mpi suites under mpiexec + all single-process reference suites
TAG=${1:-r03s16}
bash tools/r03_s15.sh $TAG/mpi
bash tools/run_reftests.sh gpurun_out/$TAG/reftests > /dev/null
tail -100 gpurun_out/$TAG/reftests/summary.txt | awk '{f+=substr($6,8)} END {print "failed total", f}'
cat gpurun_out/$TAG/mpi/summary.txt
