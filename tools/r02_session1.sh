#!/bin/bash
# round-2 GPU session 1: allocation schemes x wave order (tools/place_lab2), traffic counters
TAG=${1:-r02a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== partitions"; rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -i partition | head
for f in /sys/module/amdgpu/parameters/{vm_fragment_size,vm_block_size,vm_size,noretry,mtype_local}; do echo "$f = $(cat $f 2>/dev/null)"; done
for f in /sys/class/drm/card*/device/{current_memory_partition,current_compute_partition,mem_info_vram_total,mem_info_vram_used}; do echo "$f = $(cat $f 2>/dev/null)"; done
grep -m1 "model name" /proc/cpuinfo; nproc; numactl -H 2>/dev/null | head -12; lscpu | grep -i -E "numa|socket|thread" | head
} > $OUT/info.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 tools/place_lab2 256 8 all 6 > $OUT/lab2_all.txt 2>&1; echo "lab all rc=$?"
timeout 120 tools/place_lab2 256 8 quick 3 > $OUT/lab2_quick_p2.txt 2>&1; echo "lab quick rc=$?"
HSA_MAX_VA_ALIGN=30 timeout 120 tools/place_lab2 256 8 quick 3 > $OUT/lab2_quick_vaalign.txt 2>&1; echo "lab quick va rc=$?"
cd /tmp
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/pmc_1 -o p -- $GRAFT_REPO_ROOT/tools/place_lab2 256 3 pmc > $OUT/pmc_1.log 2>&1; echo "pmc1 rc=$?"
timeout 200 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum --kernel-trace --output-format csv -d $OUT/pmc_2 -o p -- $GRAFT_REPO_ROOT/tools/place_lab2 256 3 pmc > $OUT/pmc_2.log 2>&1; echo "pmc2 rc=$?"
cd $GRAFT_REPO_ROOT
python - $OUT <<'PY' > $OUT/pmc_summary.txt 2>&1
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    # dispatch order tells the wave order: the lab runs map 0 first, then map 1
    for r in rows:
        k = r["Kernel_Name"]
        if "pipe3" in k: k = "csr_spmv_pipe3"
        elif "jacobi_apply_fixed" in k: k = "jacobi_apply_fixed"
        else: continue
        acc[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        v.sort()
        vals = [x[1] for x in v]
        h = len(vals) // 2
        # skip the launches of measure() (first part), keep the tail = pmc loops: last 2*reps
        tail = vals[-6:]
        print(f"   {c:40s} map0 {sum(tail[:3])/3:16.0f}   map1 {sum(tail[3:])/3:16.0f}   (all n={len(vals)})")
PY
cat $OUT/pmc_summary.txt
timeout 600 python -m pytest tests/test_spmv_gpu.py -x -q 2>&1 | tail -5 | tee $OUT/pytest_spmv.txt
