#!/bin/bash
TAG=${1:-r03s31}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests/test_spmv_gpu.py -m gpu -x -q 2>&1 | tail -2
FORMATS=csr python tools/multi_rhs_bench.py 256 > $OUT/multi_rhs_256.txt 2>&1
grep "tuning\|nrhs\|rror" $OUT/multi_rhs_256.txt
