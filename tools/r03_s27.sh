#!/bin/bash
TAG=${1:-r03s27}
OUT=gpurun_out/$TAG
mkdir -p $OUT
FORMATS=csr python tools/multi_rhs_bench.py 256 2=98 2=130 2=146 2=162 > $OUT/multi_rhs_csr_frag.txt 2>&1
grep "tuning\|nrhs [1348]\|rror" $OUT/multi_rhs_csr_frag.txt
