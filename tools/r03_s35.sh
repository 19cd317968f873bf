#!/bin/bash
TAG=${1:-r03s35}
OUT=gpurun_out/$TAG
mkdir -p $OUT
SECONDS=0
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest_gpu_tail.txt
echo "full GPU suite: $SECONDS s"
python tools/format_bench.py 256 > $OUT/format_bench_256.txt 2>&1; grep "csr\|coo" $OUT/format_bench_256.txt
python tools/dtype_bench.py 256 > $OUT/dtype_bench_256.txt 2>&1; grep csr $OUT/dtype_bench_256.txt
