#!/bin/bash
# round 6, session 13: the whole GPU suite (not -x: everything that fails in one go)
OUT=gpurun_out/r06s13
mkdir -p $OUT
export TMPDIR=/tmp
timeout 3300 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $OUT/pytest_gpu_tail.txt
