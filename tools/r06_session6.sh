#!/bin/bash
# round 6, session 6: the lane = (block, row) Jacobi apply for float / complex / adaptive storage (parity + timing),
# kernel stats of the bench's CG loop (1.54 ms per iteration in session 5 against 1.42-1.44 before) and of complex CB-GMRES
OUT=gpurun_out/r06s6
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity: jacobi types"
timeout 1500 python -m pytest tests/test_jacobi_types_gpu.py tests/test_jacobi_types_fixture_cpu.py -q -x 2>&1 | tail -6 | tee $OUT/parity_jacobi.txt
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
echo "== jacobi timings (new lanes kernel, then GKOC_TUNE_15=1 = thread-per-row)"
(cd $D && timeout 600 ./round5_bench 256 30 jacobi 2>&1 | tail -7) | tee $OUT/jacobi_new.txt
(cd $D && GKOC_TUNE_15=1 timeout 600 ./round5_bench 256 30 jacobi 2>&1 | tail -7) | tee $OUT/jacobi_old.txt
echo "== bench under rocprofv3 (kernel stats)"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu --no-ginkgo-api > $GRAFT_REPO_ROOT/$OUT/bench_line_profiled.json 2> $GRAFT_REPO_ROOT/$OUT/bench_profiled.err)
cp $(find /tmp/prof_b -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv
head -16 $OUT/bench_kernel_stats.csv | cut -c1-200
tail -1 $OUT/bench_line_profiled.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cg', d.get('cg_iters_per_s'), d.get('cg_ms_per_iter'), 'gmres', d.get('gmres_ms_per_iter'), 'spmv', d.get('ms_per_step'))"
echo "== complex CB-GMRES under rocprofv3"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o c -- $D/round5_bench 256 30 cbc > $GRAFT_REPO_ROOT/$OUT/cbc.log 2>&1)
cp $(find /tmp/prof_c -name '*kernel_stats.csv' | head -1) $OUT/cbc_kernel_stats.csv
head -24 $OUT/cbc_kernel_stats.csv | cut -c1-220
tail -2 $OUT/cbc.log
