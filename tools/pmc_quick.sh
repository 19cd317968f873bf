#!/bin/bash
# one rocprofv3 --pmc pass over the lab's pmc mode; per-kernel mean counters
# usage: bash tools/pmc_quick.sh <tag> "<counters>" [reps]
TAG=$1; CNT=$2; REPS=${3:-3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/pmc_1 -o p -- $GRAFT_REPO_ROOT/tools/spmv_lab 256 $REPS pmc > $OUT/pmc_1.log 2>&1
echo "rc=$?"; grep -E "pipe|PRODUCTION" $OUT/pmc_1.log
cd $GRAFT_REPO_ROOT
python - <<'PY' $OUT | tee $OUT/pmc_quick_summary.txt
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_1", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pipe" not in k: continue
        k = k[k.index("pipe"):][:90]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:36s} {sum(v)/len(v):16.0f} (n={len(v)})")
PY
