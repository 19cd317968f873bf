#!/bin/bash
# Ginkgo's own benchmark drivers on this backend (oracle/build_benchmarks.py): 27-pt stencil with
# 128^3 = 2 097 152 rows (the host-side matrix_data assembly of the harness makes 256^3 take minutes)
OUT=${1:-gpurun_out/gko_bench}
mkdir -p $OUT
B=${GRAFT_REPO_ROOT:-.}/oracle/_ref/dropin/benchmark
echo '[{"stencil": "27pt", "size": 2097152}]' | timeout 900 $B/spmv -executor hip -formats csr,coo,ell,sellp,hybrid -gpu_timer > $OUT/spmv_27pt_128.json 2> $OUT/spmv_27pt_128.err; echo "spmv rc=$?"
echo '[{"stencil": "27pt", "size": 2097152, "optimal": {"spmv": "csr"}}]' | timeout 900 $B/solver -executor hip -solvers cg,bicgstab,gmres -preconditioners jacobi -jacobi_max_block_size 8 -max_iters 1000 -rel_res_goal 1e-10 -gpu_timer > $OUT/solver_27pt_128.json 2> $OUT/solver_27pt_128.err; echo "solver rc=$?"
python - $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/spmv_27pt_128.json"))[0]
print("rows", d["rows"], "nnz", d["nonzeros"])
for f, r in d["spmv"].items():
    print(f"  {f:8s} {r['time']*1e6:9.1f} us  storage {r['storage']}  rel.err {r['max_relative_norm2']:.1e}")
s = json.load(open(sys.argv[1] + "/solver_27pt_128.json"))[0]["solver"]
for name, r in s.items():
    print(f"  {name:10s} iterations {r['apply']['iterations']:5d}  apply {r['apply']['time']*1e3:8.2f} ms  residual {r['residual_norm']:.2e}")
PY
