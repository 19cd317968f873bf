#!/bin/bash
# round 6, session 7: after the register fix of the CSR kernel, the unconditional loads of the Jacobi lanes kernel and
# the non-blocking MPI_Test: parity on what was touched, the bench line, the timings again
OUT=gpurun_out/r06s7
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity"
timeout 2000 python -m pytest tests/test_spmv_gpu.py tests/test_flan_like_gpu.py tests/test_jacobi_types_gpu.py tests/test_krylov_gpu.py tests/test_mpi_dropin_gpu.py tests/test_coo_gpu.py -m gpu -q -x 2>&1 | tail -8 | tee $OUT/parity.txt
echo "== bench (default command)"
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "rc=$?"
tail -1 $OUT/bench_line.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('value',d['value'],'frac',r['frac'],'cg',d.get('cg_iters_per_s'),d.get('cg_ms_per_iter'),'gmres',d.get('gmres_iters_per_s'),'api',d.get('ginkgo_api',{}).get('cg_iters_per_s'), d.get('ginkgo_api',{}).get('one_kernel_per_call'))
"
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
echo "== jacobi timings"
(cd $D && timeout 600 ./round5_bench 256 30 jacobi 2>&1 | tail -7) | tee $OUT/jacobi_new.txt
echo "== irregular + formats + flan + dtype"
timeout 300 python tools/irregular_pmc.py 2>&1 | tail -1 | tee $OUT/irregular.txt
timeout 600 python tools/format_bench.py 256 2>&1 | tail -8 | tee $OUT/formats.txt
timeout 600 python tools/flan_bench.py 80 2>&1 | tail -5 | tee $OUT/flan.txt
