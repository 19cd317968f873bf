#!/bin/bash
# round 2, session 6: native distributed driver, distributed PipeCg, config-3 slab test, host cost,
# PMC comparison CSR vs SELL-P on the Flan-like matrix
TAG=${1:-r02s6}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_native_cg_gpu.py tests/test_distributed.py tests/test_krylov_family_gpu.py -q -x -m gpu 2>&1 | tail -15
timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -x -k "config3" 2>&1 | tail -5
echo "== host cost of an iteration (mirror, 16^3: the device never limits)"
for sv in cg pipe_cg; do
  examples/native_dist_cg 16 3000 1e-30 $sv 4 mirror | tee -a $OUT/native_host_cost.txt
done
examples/native_dist_cg 256 100 1e-30 cg 4 | tee -a $OUT/native_l256.txt
examples/native_dist_cg 256 100 1e-30 pipe_cg 4 | tee -a $OUT/native_l256.txt
timeout 300 python tools/dist_host_cost.py 16 400 direct 2>&1 | tail -12 | tee $OUT/python_host_cost.txt
echo "== bench through the distributed path with PipeCg"
GKO_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu --pipe-cg > $OUT/bench_forcedist_pipe.json 2> $OUT/bench_forcedist_pipe.err; echo "rc=$?"; tail -c 900 $OUT/bench_forcedist_pipe.json; tail -3 $OUT/bench_forcedist_pipe.err
echo "== PMC: CSR vs SELL-P on the Flan-like matrix"
cd /tmp
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- python $GRAFT_REPO_ROOT/tools/flan_pmc.py 80 3 > $OUT/pmc_$i.log 2>&1
  echo "pass $i: $GROUP -> rc=$?"
done <<'GROUPS'
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_HIT_sum TCC_MISS_sum
GROUPS
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT 2>/dev/null | grep -A30 -E "^csr_spmv_pipe3|^sellp_spmv_kernel" | tee $OUT/pmc_summary.txt | head -80
