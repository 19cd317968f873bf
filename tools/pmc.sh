#!/bin/bash
# PMC passes (one rocprofv3 run per counter group, kernel-trace only) over a
# command; summaries land in gpurun_out/<tag>/pmc_<i>/.  usage:
#   bash tools/pmc.sh <tag> <command...>
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- "$@" > $OUT/pmc_$i.log 2>&1
  echo "pass $i: $GROUP -> rc=$?"
done <<'GROUPS'
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_LATENCY_sum
GRBM_GUI_ACTIVE TCP_TCP_LATENCY_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum
GROUPS
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT | tee $OUT/pmc_summary.txt
