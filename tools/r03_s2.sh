#!/bin/bash
TAG=${1:-r03s2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== 8 ranks, 64^3, worker directly"
OMP_NUM_THREADS=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29656 tests/dist_worker.py gpu 64 > $OUT/w8_64.txt 2>&1; echo "rc=$?"
grep -v "Gloo\|amdgpu.ids" $OUT/w8_64.txt | grep -B2 -A25 "Traceback" | head -80
grep "dist_worker OK" $OUT/w8_64.txt
echo "== bench path with 8 ranks (gloo)"
timeout 900 python -m pytest tests/test_distributed.py -m gpu -q -x -k "bench_command or fallback" 2>&1 | tail -30
echo "== full-size parity vs reference omp"
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -s -k "reference_omp" 2>&1 | tail -15 | tee $OUT/fullsize_ref.txt
exit 0
