#!/bin/bash
OUT=gpurun_out/r06s30
mkdir -p $OUT
export TMPDIR=/tmp
LAYS=0,5,6 timeout 900 python tools/layout_ab.py 2>&1 | grep "layout" | tee $OUT/layout_ring16k.txt
