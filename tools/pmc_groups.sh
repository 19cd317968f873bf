#!/bin/bash
# usage: bash tools/pmc_groups.sh <tag> <kernel-name regex> -- <command ...>
# One rocprofv3 pass with --kernel-trace --stats, then one --pmc pass per counter group (kernel-trace only, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes) over the SAME command; per-kernel means of every counter
# and the kernel durations go to gpurun_out/<tag>/summary.txt.
TAG=$1; RE=$2; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1
echo "trace rc=$?"; tail -2 $OUT/trace.log
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- "$@" > $OUT/pmc_$i.log 2>&1
  echo "pass $i: $GROUP -> rc=$?"
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE
TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum
TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_REQ_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
SQ_ACCUM_PREV_HIRES SQ_LEVEL_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
GROUPS
cd $GRAFT_REPO_ROOT
python - "$OUT" "$RE" <<'PY' | tee $OUT/summary.txt
import csv, glob, os, re, sys
from collections import defaultdict
root, rx = sys.argv[1], re.compile(sys.argv[2])
def short(k):
    if not rx.search(k): return None
    k = re.sub(r"\(.*$", "", k)
    return k[:150]
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        s = short(r.get("Kernel_Name", ""))
        if s: dur[s].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel durations (us), trace pass: launches / min / mean / max")
for k, v in dur.items():
    print(f"  {k}\n      {len(v)} / {min(v):.1f} / {sum(v)/len(v):.1f} / {max(v):.1f}")
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
