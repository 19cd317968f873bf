#!/bin/bash
# round 6, session 11: the drop-in's CG after the polling fix and with csr::spmv + <p,q>; irregular with the XCD
# rule and the pipelined chunk loop; parity of the touched paths
OUT=gpurun_out/r06s11
mkdir -p $OUT
export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/oracle/_ref/dropin
export LD_LIBRARY_PATH=$D:$D/../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib
echo "== parity"
timeout 2400 python -m pytest tests/test_spmv_gpu.py tests/test_flan_like_gpu.py tests/test_complex_gpu.py tests/test_dropin_gpu.py -m gpu -q 2>&1 | tail -6 | tee $OUT/parity.txt
(cd $D && timeout 900 ./dropin_test 2>&1 | grep -i "FAIL\|passed\|failed\|anticipated:\|by-products:" | head -12) | tee $OUT/dropin_lines.txt
echo "== api gap"
bash tools/api_gap.sh r06s11/api_gap > /dev/null 2>&1
sed -n 1,40p $OUT/api_gap/report.txt
tail -2 $OUT/api_gap/plain.txt | cut -c1-400
rm -rf $OUT/api_gap/trace
echo "== irregular"
timeout 300 python tools/irregular_pmc.py 2>&1 | tail -1 | tee $OUT/irregular.txt
echo "== bench"
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "rc=$?"
tail -1 $OUT/bench_line.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('value',d['value'],'frac',r['frac'],'cg',d.get('cg_iters_per_s'),d.get('cg_ms_per_iter'),'gmres',d.get('gmres_iters_per_s'),'api',d.get('ginkgo_api',{}).get('cg_iters_per_s'), d.get('ginkgo_api',{}).get('one_kernel_per_call'))
"
