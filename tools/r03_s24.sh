#!/bin/bash
TAG=${1:-r03s24}
OUT=gpurun_out/$TAG
mkdir -p $OUT
FORMATS=ell python tools/multi_rhs_bench.py 256 2=2,6=0 2=2,6=1024 2=2,6=2048 2=2,6=4096 2=2,6=8192 2=2,6=16384 2=2,6=32768 > $OUT/multi_rhs_frag_chunk.txt 2>&1
grep "tuning\|nrhs [248]" $OUT/multi_rhs_frag_chunk.txt
