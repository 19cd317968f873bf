#!/bin/bash
OUT=gpurun_out/r06s28
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/layout_ab.py 2>&1 | grep layout | tee $OUT/layout_ab.txt
