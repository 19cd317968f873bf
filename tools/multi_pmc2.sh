#!/bin/bash
# usage: bash tools/multi_pmc2.sh <tag>   - what bounds the CSR SpMV with 4 / 8 right-hand sides:
# instruction mix, LDS waits and bank conflicts, L1 stalls, memory latency, occupancy.  One rocprofv3 --pmc pass per group.
TAG=${1:-multi_pmc2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp NOELL=1 NRHS=${NRHS:-1,4,8}
cd /tmp
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- python $GRAFT_REPO_ROOT/tools/multi_pmc.py > $OUT/pmc_$i.log 2>&1
  echo "pass $i: $GROUP -> rc=$?"
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE
TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum
SQ_ACCUM_PREV_HIRES SQ_LEVEL_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
GROUPS
cd $GRAFT_REPO_ROOT
python - <<'PY' $OUT | tee $OUT/multi_pmc_summary.txt
import csv, glob, os, re, sys
from collections import defaultdict
root = sys.argv[1]
def short(k):
    m = re.search(r"(csr_spmv_frag\w*|csr_spmv_multi_kernel|csr_spmv_rowmulti_kernel|csr_spmv_pipe3_kernel)<([^>]*)>", k)
    if not m: return None
    return m.group(1) + "<" + m.group(2).replace("double, int, ", "") + ">"
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d): continue
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            s = short(r.get("Kernel_Name", ""))
            if s: per[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            s = short(r.get("Kernel_Name", ""))
            if s: dur[s].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(os.path.basename(d))
    for k in per:
        print("  ", k, f"  [{min(dur[k]):.0f} us]" if dur.get(k) else "")
        for c, v in sorted(per[k].items()):
            print(f"      {c:50s} {sum(v)/len(v):18.0f}  ({len(v)} launches)")
PY
