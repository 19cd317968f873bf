#!/bin/bash
OUT=gpurun_out/r06s29
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_spmv_gpu.py tests/test_flan_like_gpu.py tests/test_mixed_gpu.py tests/test_fullsize_gpu.py -m gpu -q 2>&1 | tail -4 | tee $OUT/parity.txt
timeout 900 python tools/layout_ab.py 2>&1 | grep "f32" | tee $OUT/layout_ab_after.txt
