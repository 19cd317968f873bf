#!/bin/bash
# round 6, session 25: eight columns with eight lanes per row (8-row segments, 3 KB of LDS per wave) against the default
OUT=gpurun_out/r06s25
mkdir -p $OUT
export TMPDIR=/tmp
FORMATS=csr timeout 900 python tools/multi_rhs_bench.py 256 11=0 11=8040 11=8080 11=8160 11=9040 11=9080 2>&1 | grep -v amdgpu | tee $OUT/multi_rhs_8rows.txt
