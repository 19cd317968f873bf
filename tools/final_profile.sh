#!/bin/bash
# Round-end evidence: kernel-trace stats of the default bench + HBM traffic of the
# SpMV kernel from PMC passes (one counter group per rocprofv3 run, kernel-trace
# only).  usage (repo root, GPU box): bash tools/final_profile.sh <tag>
TAG=${1:-final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# the bench line and the kernel-trace statistics come from ONE process (the
# default bench command, minus the CPU baseline whose 256 OpenMP threads crawl
# under the tracer, under rocprofv3 --kernel-trace --stats), so the kernel's
# average duration in the trace and bench.py's own HIP-event figure describe
# the same launches (different processes land on different placements, 3.2)
echo "== default bench command under rocprofv3 --kernel-trace --stats"
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu > $OUT/trace_run.txt 2>&1
grep '^{"metric"' $OUT/trace_run.txt | tail -1 | tee $OUT/bench_line.json
find $OUT/trace -name "*kernel_stats*" | head -1 | xargs -r head -12
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cg-iters 0 --no-cpu --placement 0 > $OUT/pmc_$i.log 2>&1
  echo "pmc pass $i: $GROUP -> rc=$?"
done <<'GROUPS'
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
FETCH_SIZE
WRITE_SIZE
GROUPS
cd $GRAFT_REPO_ROOT
python - $OUT <<'PY' | tee $OUT/spmv_pmc.json
import csv, glob, json, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(list)
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "csr_spmv_pipe3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
rd = m.get("TCC_EA0_RDREQ_128B_sum", 0) * 128 + m.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + m.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
wr64 = m.get("TCC_EA0_WRREQ_64B_sum", 0)
wr = wr64 * 64 + (m.get("TCC_EA0_WRREQ_sum", 0) - wr64) * 32
print(json.dumps({
    "kernel": "csr_spmv_pipe3_kernel<double,int,false,64,4,1,1024,1,0x2000> (production), 27-pt 256^3",
    "method": "rocprofv3 --pmc, one counter group per run, mean per dispatch; read bytes = "
              "RDREQ_128B*128 + RDREQ_64B*64 + RDREQ_32B*32, write bytes = WRREQ_64B*64 + other*32; "
              "FETCH_SIZE (KB) under-reports 128-B requests by 2x on gfx950 (MI355X_MICROARCH.md), shown for reference",
    "counters_mean_per_launch": m,
    "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr),
    "hbm_bytes_per_launch": int(rd + wr)}, indent=1))
PY
