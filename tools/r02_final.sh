#!/bin/bash
# Round-2 evidence run on one MI355X box: the full GPU test-suite, the default bench line, the same
# command under rocprofv3 --kernel-trace --stats, HBM-traffic counters of the SpMV kernel (one
# counter group per rocprofv3 run, kernel-trace only), Ginkgo's API on this backend, the Flan-like
# stand-in, the native distributed driver.  usage (repo root, GPU box): bash tools/r02_final.sh <tag>
TAG=${1:-r02final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{ rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx9" | head -6; echo "nproc: $(nproc)"; grep -m1 "model name" /proc/cpuinfo; } > $OUT/box_info.txt 2>&1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 | tee $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
echo "== default bench command"
timeout 900 python bench.py 2> $OUT/bench_default.err | grep '^{"metric"' | tail -1 | tee $OUT/bench_line_unprofiled.json | cut -c1-400
echo "== default bench command under rocprofv3 --kernel-trace --stats"
(cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu > $OUT/trace_run.txt 2>&1)
grep '^{"metric"' $OUT/trace_run.txt | tail -1 > $OUT/bench_line_profiled.json
find $OUT/trace -name "*kernel_stats*" | head -1 | xargs -r -I{} cp {} $OUT/bench_kernel_stats.csv
head -9 $OUT/bench_kernel_stats.csv | cut -c1-230
rm -rf $OUT/trace
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cg-iters 0 --no-cpu > $OUT/pmc_$i.log 2>&1)
  echo "pmc pass $i: $GROUP -> rc=$?"
done <<'GROUPS'
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
FETCH_SIZE
WRITE_SIZE
GROUPS
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
rm -rf $OUT/pmc_*/
echo "== Ginkgo's own API on this backend (dropin_bench 256 50 100)"
(cd oracle/_ref/dropin && timeout 600 ./dropin_bench 256 50 100 2>&1 | tee $OUT/ginkgo_api_bench.txt | tail -12)
echo "== Flan-like stand-in"
timeout 600 python tools/flan_bench.py 80 2>&1 | grep -E "flan-like|SpMV|it/s" | tee $OUT/flan_like.txt
echo "== native distributed driver"
for sv in cg pipe_cg; do
  GKOC_EXAMPLE_TRACE=1 examples/native_dist_cg 16 3000 1e-30 $sv 8 mirror 2>&1 | grep -E "^\{|host us" | tee -a $OUT/native_dist.txt
done
GKOC_EXAMPLE_TRACE=1 examples/native_dist_cg 16 3000 1e-30 cg 8 2>&1 | grep -E "^\{|host us" | tee -a $OUT/native_dist.txt
GKOC_EXAMPLE_NULL_STREAM=1 GKOC_EXAMPLE_TRACE=1 examples/native_dist_cg 16 3000 1e-30 cg 8 2>&1 | grep -E "^\{|host us" | sed "s/^/null stream: /" | tee -a $OUT/native_dist.txt
examples/native_dist_cg 256 100 1e-30 cg 4 | grep "^{" | tee -a $OUT/native_dist.txt
examples/native_dist_cg 256 100 1e-30 pipe_cg 4 | grep "^{" | tee -a $OUT/native_dist.txt
echo "== all formats"
timeout 600 python tools/format_bench.py 256 2>&1 | grep -v amdgpu.ids | tee $OUT/all_formats_256.txt
