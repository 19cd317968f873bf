#!/bin/bash
TAG=${1:-r03s23}
OUT=gpurun_out/$TAG
mkdir -p $OUT
FORMATS=ell,sellp python tools/multi_rhs_bench.py 256 2=1 2=2 > $OUT/multi_rhs_frag2.txt 2>&1
grep "tuning\|nrhs" $OUT/multi_rhs_frag2.txt
python -m pytest tests -m gpu -x -q -k "ell or sellp or format or multi" 2>&1 | tail -3
