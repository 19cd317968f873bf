#!/bin/bash
# multi-RHS SpMV: XCD chunk sweep
TAG=${1:-r03s19}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python tools/multi_rhs_bench.py 256 0 1024 2048 4096 8192 16384 32768 > $OUT/multi_rhs_chunk_sweep.txt 2>&1
cat $OUT/multi_rhs_chunk_sweep.txt
