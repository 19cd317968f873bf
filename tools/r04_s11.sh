#!/bin/bash
# round 4, session 11: configs[4] as a bench configuration (stand-in at Flan scale, 1 rank; 8 gloo ranks small)
TAG=${1:-r04s11}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"
timeout 1500 python -m pytest tests/test_distributed.py -m gpu -q -x -k "configs4 or eight_ranks_one_device or two_ranks or three_ranks" 2>&1 | tail -12 | tee $OUT/tests.txt
echo "== stand-in at Flan scale, one GPU"
for f in csr sellp; do
timeout 900 python bench.py --workload flan --format $f --steps 20 --warmup 5 --cg-iters 100 > $OUT/flan_$f.json 2> $OUT/flan_$f.err; tail -1 $OUT/flan_$f.json | cut -c1-1800
done
echo "== 8 gloo ranks on this GPU at Flan scale (numbers mean nothing: host-staged, one device; the path runs)"
GKO_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 8 --workload flan --format sellp --steps 5 --warmup 2 --cg-iters 20 > $OUT/flan_8gloo.json 2> $OUT/flan_8gloo.err; tail -1 $OUT/flan_8gloo.json | cut -c1-2500
tail -5 $OUT/flan_8gloo.err
echo done
