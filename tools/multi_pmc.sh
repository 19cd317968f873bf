#!/bin/bash
# usage: bash tools/multi_pmc.sh <tag>   (development tool: counters of the multi-RHS SpMV kernels)
TAG=${1:-multi_pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- python $GRAFT_REPO_ROOT/tools/multi_pmc.py > $OUT/pmc_$i.log 2>&1
  echo "pass $i: $GROUP -> rc=$?"
done <<'GROUPS'
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_TOTAL_CACHE_ACCESSES_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD
TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum
GROUPS
cd $GRAFT_REPO_ROOT
python - <<'PY' $OUT | tee $OUT/multi_pmc_summary.txt
import csv, glob, os, re, sys
from collections import defaultdict
root = sys.argv[1]
def short(k):
    m = re.search(r"(fmt_spmv_multi_kernel|fmt_spmv_frag_kernel|csr_spmv_frag_kernel|csr_spmv_multi_kernel|csr_spmv_rowmulti_kernel|ell_spmv\w*|csr_spmv_pipe3_kernel)<([^>]*)>", k)
    if not m: return None
    return m.group(1) + "<" + m.group(2).replace("double, int, ", "") + ">"
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d): continue
    per = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            s = short(r.get("Kernel_Name", ""))
            if s: per[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d))
    for k in per:
        print("  ", k)
        for c, v in sorted(per[k].items()):
            print(f"      {c:50s} {sum(v)/len(v):18.0f}  ({len(v)} launches)")
PY
