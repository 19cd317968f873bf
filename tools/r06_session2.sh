#!/bin/bash
# round 6, second session: parity of the new kernels, then their timings
OUT=gpurun_out/r06s2
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity"
timeout 1200 python -m pytest tests/test_spmv_gpu.py tests/test_flan_like_gpu.py -m gpu -q -x 2>&1 | tail -15 | tee $OUT/parity.txt
echo "== irregular"
timeout 300 python tools/irregular_pmc.py 2>&1 | tail -2
for S in 1 2 4 8; do TUNE=13=$S timeout 300 python tools/irregular_pmc.py 2>&1 | tail -1 | sed "s/^/spw=$S: /"; done
echo "== irregular trace"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/irr_trace -o t -- python $GRAFT_REPO_ROOT/tools/irregular_pmc.py > $GRAFT_REPO_ROOT/$OUT/irr_trace.log 2>&1)
f=$(find $OUT/irr_trace -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-200
echo "== multi rhs: default, then the neighbour-reuse variants"
FORMATS=csr timeout 900 python tools/multi_rhs_bench.py 256 11=0 11=5040 11=6040 11=7040 11=5020 11=5080 2>&1 | tail -40 | tee $OUT/multi_rhs.txt
echo "== headline again (the kernel now walks runs)"
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu --no-ginkgo-api --no-pmc --gmres-iters 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value',d['value'],'frac',r['frac'],'peak_measured',r.get('peak_measured'),r.get('frac_of_measured'),r.get('peak_measured_how'),'cg',d.get('cg_iters_per_s'))"
echo "== 5-pt 4096^2 and small sizes (the segs-per-wave rule must not cost them)"
timeout 600 python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, ctypes as C
import ginkgo_amd as g
ex = g.Cdna4Executor.create(0)
def t(a, reps=20):
    n = a.size[0]
    x = g.Dense.from_numpy(ex, np.random.default_rng(1).uniform(-1, 1, n)); y = g.Dense.create(ex, (n, 1))
    for _ in range(5): a.apply(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): a.apply(x, y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, mk in (("5pt 4096^2", lambda: g.stencil_csr(ex, 2, 4096, restricted=True)), ("5pt 2048^2", lambda: g.stencil_csr(ex, 2, 2048, restricted=True)),
                 ("7pt? 27pt 128^3", lambda: g.stencil_csr(ex, 3, 128)), ("27pt 64^3", lambda: g.stencil_csr(ex, 3, 64))):
    a = mk()
    row = []
    for spw in (0, 1, 2, 4, 8):
        g._lib.lib().gkoc_tune_set(C.c_int(13), C.c_int64(spw))
        row.append(f"spw={spw}: {t(a):.1f} us")
    g._lib.lib().gkoc_tune_set(C.c_int(13), C.c_int64(0))
    print(name, " | ".join(row))
PY
