#!/bin/bash
TAG=${1:-r03s21}
OUT=gpurun_out/$TAG
mkdir -p $OUT
FORMATS=ell,sellp python tools/multi_rhs_bench.py 256 2=1,6=0 2=2,6=0 2=18,6=0 2=34,6=0 2=66,6=0 2=34,6=4096 2=34,6=8192 > $OUT/multi_rhs_frag.txt 2>&1
cat $OUT/multi_rhs_frag.txt
