#!/bin/bash
# round 6, session 12: hub rows beside the main kernel; then the whole GPU suite
OUT=gpurun_out/r06s12
mkdir -p $OUT
export TMPDIR=/tmp
echo "== irregular: hubs beside (default) / behind (GKOC_TUNE_17=1)"
timeout 300 python tools/irregular_pmc.py 2>&1 | tail -1 | tee $OUT/irregular.txt
TUNE=17=1 timeout 300 python tools/irregular_pmc.py 2>&1 | tail -1 | tee -a $OUT/irregular.txt
echo "== whole GPU suite"
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 | tee $OUT/pytest_gpu_tail.txt
