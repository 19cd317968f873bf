#!/bin/bash
# The whole GPU suite with its durations (where the minutes of `pytest -m gpu` go): gpurun_out/<tag>/durations.txt
# About 16 GPU-minutes at the end of round 4 (24 before ranks sharing a device took turns surveying it).
TAG=${1:-suite}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu --durations=40 2>&1 | tail -60 | tee $OUT/durations.txt
