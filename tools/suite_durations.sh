#!/bin/bash
# The whole GPU suite as the driver runs it (-x), with its durations: gpurun_out/<tag>/pytest_gpu.txt
# (copied to profiles/rNN_pytest_gpu_tail.txt).  Usage: tools/suite_durations.sh <tag> [extra pytest args]
TAG=${1:-suite}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
START=$(date +%s)
timeout 2400 python -m pytest tests/ -q -m gpu --durations=60 "$@" > $OUT/pytest_gpu_full.txt 2>&1
echo "exit code $? after $(( $(date +%s) - START )) s (the driver's step limit is 1200 s)" >> $OUT/pytest_gpu_full.txt
tail -90 $OUT/pytest_gpu_full.txt | tee $OUT/pytest_gpu.txt
