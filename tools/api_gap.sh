#!/bin/bash
# Where do the microseconds between the native C-ABI path and the unmodified Ginkgo core on the drop-in go
# ?  kernel-trace + hip-trace of tests/dropin/dropin_bench.cpp: per-kernel
# durations against gko::Timer's time per apply / per CG iteration, the idle time between kernels, and the
# HIP calls the core makes.   gpurun_out/<tag>/
TAG=${1:-api_gap}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
EXE=$PWD/oracle/_ref/dropin/dropin_bench
ROOT=$PWD
cd oracle/_ref/dropin
$EXE 256 50 200 --json > $OUT/plain.txt 2>&1
rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $OUT/trace -o t -- $EXE 256 50 200 --json > $OUT/traced.txt 2>&1
cd $OUT
python $ROOT/tools/api_gap_report.py trace > report.txt 2>&1
find trace -name "*hip_api_trace.csv" -size +64M -delete
