#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprof summary, kernel lab.
# usage (from the repo root on the GPU box): bash tools/gpu_session.sh [tag] [steps...]
TAG=${1:-r01}; shift
STEPS=${@:-"info pytest smoke bench lab prof"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for S in $STEPS; do
case $S in
info)
  echo "== rocminfo"; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx9" | head -6
  rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -i "partition" | head -4
  echo "== nproc: $(nproc)"; grep -m1 "model name" /proc/cpuinfo ;;
pytest)
  echo "== pytest -m gpu"
  timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt ;;
smoke)
  echo "== smoke"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt ;;
bench)
  echo "== bench"
  timeout 900 python bench.py --steps 30 --warmup 5 2>&1 | tail -3 | tee $OUT/bench.txt ;;
lab)
  echo "== lab"
  timeout 600 tools/spmv_lab 256 20 2>&1 | tee $OUT/lab.txt ;;
prof)
  echo "== rocprofv3 kernel-trace"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --cg-iters 20 --no-cpu > $GRAFT_REPO_ROOT/$OUT/prof_run.txt 2>&1)
  find $OUT/prof -type f | head -20
  find $OUT/prof -name "*kernel_stats*" | head -1 | xargs -r head -25 ;;
esac
done
