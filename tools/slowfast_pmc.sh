#!/bin/bash
# usage: bash tools/slowfast_pmc.sh <tag>   (development tool)
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- python $GRAFT_REPO_ROOT/tools/slowfast_pmc.py > $OUT/pmc_$i.log 2>&1
  echo "pass $i: $GROUP -> rc=$?"; grep -E "slow y" $OUT/pmc_$i.log
done <<'GROUPS'
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum
TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum
GROUPS
cd $GRAFT_REPO_ROOT
python - <<'PY' $OUT | tee $OUT/slowfast_summary.txt
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d): continue
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "csr_spmv" in r.get("Kernel_Name", "")]
    per = defaultdict(dict)
    for r in rows:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)[-10:]
    names = sorted({c for i in ids for c in per[i]})
    print(os.path.basename(d))
    for c in names:
        s = [per[i].get(c, 0) for i in ids[:5]]; f_ = [per[i].get(c, 0) for i in ids[5:]]
        print(f"   {c:44s} slow {sum(s)/5:16.0f}   fast {sum(f_)/5:16.0f}   ratio {sum(s)/max(sum(f_),1):.3f}")
PY
