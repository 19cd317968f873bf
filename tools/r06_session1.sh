#!/bin/bash
# round 6, first GPU session: the new bench line, the first-contact fault tests, baselines + counters for the
# kernels this round works on (irregular CSR, multi-column CSR, the 55-66 % cluster)
OUT=gpurun_out/r06s1
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench (default command)"
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "rc=$?"
tail -1 $OUT/bench_line.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('value',d['value'],'frac',r['frac'],'peak_measured',r.get('peak_measured'),'frac_of_measured',r.get('frac_of_measured'),'triad',r.get('triad_measured'))
print('cg',d.get('cg_iters_per_s'),'cg_frac',d.get('cg_frac'),'gmres',d.get('gmres_iters_per_s'),d.get('gmres_ms_per_iter'),'gmres_frac',d.get('gmres_frac'),d.get('gmres_model_frac'),d.get('gmres_error'))
print('placement',d['placement'])
print('startup_s',d.get('startup_s'),'api',d.get('ginkgo_api',{}).get('cg_iters_per_s'),d.get('ginkgo_api',{}).get('frac'))
"
tail -5 $OUT/bench.err
echo "== fault tests"
timeout 1500 python -m pytest tests/test_distributed.py -m gpu -q -x -k "survives or single_memory_class or other_devices or device_resident_transport" 2>&1 | tail -15 | tee $OUT/fault_tests.txt
echo "== irregular: trace + counters"
bash tools/pmc_groups.sh r06s1/irr 'csr_spmv_pipe3|csr_flagged|csr_long' -- python $GRAFT_REPO_ROOT/tools/irregular_pmc.py > $OUT/irr_pmc.log 2>&1
head -12 $OUT/irr/summary.txt; grep -h "irregular n=" $OUT/irr/*.log | head -2
echo "== multi rhs baseline"
FORMATS=csr timeout 600 python tools/multi_rhs_bench.py 256 2>&1 | tail -8 | tee $OUT/multi_rhs.txt
echo "== formats baseline"
timeout 600 python tools/format_bench.py 256 2>&1 | tail -12 | tee $OUT/formats.txt
timeout 600 python tools/flan_bench.py 80 2>&1 | tail -8 | tee $OUT/flan.txt
timeout 600 python tools/dtype_bench.py 2>&1 | tail -16 | tee $OUT/dtype.txt
