#!/bin/bash
OUT=gpurun_out/${1:-r03s45}
mkdir -p $OUT
python -m pytest tests/test_jacobi_mfma_gpu.py -m gpu -x -q 2>&1 | tail -5
python tools/jacobi_mfma_bench.py 256 > $OUT/jacobi_multi_256.txt 2>&1; cat $OUT/jacobi_multi_256.txt
python -m pytest tests -m gpu -x -q -k "jacobi or reference_suite" 2>&1 | tail -3
