#!/bin/bash
TAG=${1:-r03s22}
OUT=gpurun_out/$TAG
mkdir -p $OUT
FORMATS=ell python tools/multi_rhs_bench.py 256 2=2,6=0 2=2,6=2048 2=2,6=8192 2=2,6=16384 > $OUT/multi_rhs_frag_chunk.txt 2>&1
grep "tuning\|nrhs [48]" $OUT/multi_rhs_frag_chunk.txt
TUNE=2=2 NOCSR=1 bash tools/multi_pmc.sh $TAG/pmc > /dev/null 2>&1
cat $OUT/pmc/multi_pmc_summary.txt
