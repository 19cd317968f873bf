#!/bin/bash
# First contact with a multi-GPU node: nothing here has ever run on more than one
# device.  In order: the communicators' own checks at 2 ranks, then bench.py at 2, 4, 8 GPUs.  Every step has a
# time limit and leaves its log under profiles/first_contact/ - a step that fails says so and the next one runs.
#   bash tools/first_contact_8gpu.sh [grid=256] [max_gpus=8]
GRID=${1:-256}
MAXG=${2:-8}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/profiles/first_contact
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=${TMPDIR:-/tmp} OMP_NUM_THREADS=1
cd "$ROOT"
NDEV=$(python -c "import torch; print(torch.cuda.device_count())")
echo "devices visible: $NDEV" | tee "$OUT/00_devices.txt"
rocm-smi --showtopo 2>/dev/null | head -60 >> "$OUT/00_devices.txt"
run() {   # run <name> <seconds> <command...>
  local name=$1 limit=$2; shift 2
  echo "== $name"
  timeout "$limit" "$@" > "$OUT/$name.out" 2> "$OUT/$name.err"
  local rc=$?
  echo "   rc=$rc  ($(tail -c 300 "$OUT/$name.out" | tr '\n' ' ' | cut -c1-200))"
  return $rc
}
launch() { python -m torch.distributed.run --nnodes=1 --nproc-per-node "$1" --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) "${@:2}"; }
# 1. each transport alone, two ranks, the smallest problem: does it come up, are the answers right, what does an
#    iteration's communication cost (comm_check in the line), who saw whom (topology)
for T in ipc rccl torch; do
  GKO_COMM=$T run "01_two_ranks_$T" 600 launch 2 bench.py --gpus 2 --grid 64 --steps 5 --warmup 2 --cg-iters 20
done
# 2. the default choice (both transports timed, the faster one taken) with the product's soak, at the real size
for N in 2 4 8; do
  [ "$N" -le "$NDEV" ] && [ "$N" -le "$MAXG" ] || continue
  run "02_bench_${N}gpus" 1200 launch "$N" bench.py --gpus "$N" --grid "$GRID" --steps 20 --warmup 5
  # the conservative shape next to it: RCCL, join-based product (nothing waits inside a kernel)
  GKO_COMM=rccl GKO_GATED_SPMV=0 run "03_bench_${N}gpus_rccl_join" 1200 launch "$N" bench.py --gpus "$N" --grid "$GRID" --steps 20 --warmup 5
done
# 3. the C++ driver on the C ABI alone (examples/native_dist_cg.cpp) where it has been built
[ -x examples/native_dist_cg ] && for N in 2 8; do
  [ "$N" -le "$NDEV" ] && [ "$N" -le "$MAXG" ] || continue
  run "04_native_dist_cg_${N}_rccl" 600 python -m torch.distributed.run --no-python --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) examples/native_dist_cg "$GRID" 100 1e-30 cg
  run "04_native_dist_cg_${N}_ipc" 600 python -m torch.distributed.run --no-python --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) examples/native_dist_cg "$GRID" 100 1e-30 pipe_cg 4 ipc
done
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "0[123]_*.out"))):
    line = [ln for ln in open(f) if ln.startswith('{"metric"')]
    if not line:
        print(os.path.basename(f), ": NO LINE"); continue
    d = json.loads(line[-1]); cc = d.get("comm_check", {}); topo = cc.get("topology", {})
    print(os.path.basename(f), "| value", d.get("value"), "| cg it/s", d.get("cg_iters_per_s"), "| pipe", d.get("pipe_cg_iters_per_s"),
          "|", cc.get("communicator"), "ranks_seen", topo.get("ranks_seen"), "rccl", topo.get("rccl_version"),
          "| all-reduce us", cc.get("all_reduce_us"), "exchange us", cc.get("exchange_us"),
          "|", d.get("distributed_product", {}).get("gate_fence"), "|", d.get("error", ""))
PY
