#!/bin/bash
TAG=${1:-r03s34}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests/test_spmv_gpu.py tests/test_coo_hybrid_gpu.py tests/test_krylov_gpu.py tests/test_cg_gpu.py tests/test_distributed.py -m gpu -x -q 2>&1 | tail -3
python tools/format_bench.py 256 > $OUT/format_bench_256.txt 2>&1; cat $OUT/format_bench_256.txt
python tools/dtype_bench.py 256 > $OUT/dtype_bench_256.txt 2>&1; cat $OUT/dtype_bench_256.txt
python bench.py --no-pmc --no-cpu --no-ginkgo-api --steps 50 --warmup 20 > $OUT/bench.txt 2>&1; tail -1 $OUT/bench.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'], d.get('cg_iters_per_s'))"
