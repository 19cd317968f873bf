#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprof summary, kernel lab.
# usage (from the repo root on the GPU box): bash tools/gpu_session.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx9" | head -6
echo "== nproc: $(nproc)"; grep -m1 "model name" /proc/cpuinfo
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt
echo "== bench"
timeout 900 python bench.py --steps 30 --warmup 5 2>&1 | tail -3 | tee $OUT/bench.txt
echo "== lab"
timeout 600 tools/spmv_lab 256 20 2>&1 | tee $OUT/lab.txt
echo "== rocprofv3 kernel-trace"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --cg-iters 20 --no-cpu > $GRAFT_REPO_ROOT/$OUT/prof_run.txt 2>&1)
ls -R $OUT/prof | head -20
find $OUT/prof -name "*kernel_stats*" | head -1 | xargs -r head -15
