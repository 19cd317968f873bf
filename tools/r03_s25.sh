#!/bin/bash
TAG=${1:-r03s25}
OUT=gpurun_out/$TAG
mkdir -p $OUT
TUNE=2=2,6=0 NOCSR=1 bash tools/multi_pmc.sh $TAG/pmc_chunk0 > /dev/null 2>&1
TUNE=2=2,6=8192 NOCSR=1 bash tools/multi_pmc.sh $TAG/pmc_chunk8192 > /dev/null 2>&1
for d in pmc_chunk0 pmc_chunk8192; do echo "== $d"; grep -A40 "^pmc_2" $OUT/$d/multi_pmc_summary.txt | grep -B1 -A4 "frag_kernel<false, [48]" | grep "frag\|RDREQ_sum\|HIT\|MISS"; done
