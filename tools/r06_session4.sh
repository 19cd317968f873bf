#!/bin/bash
OUT=gpurun_out/r06s4
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity (spmv, flan, dropin incl. the two-thread test)"
timeout 1500 python -m pytest tests/test_spmv_gpu.py tests/test_flan_like_gpu.py tests/test_dropin_gpu.py -m gpu -q -x 2>&1 | tail -8 | tee $OUT/parity.txt
(cd oracle/_ref/dropin && LD_LIBRARY_PATH=.:../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib timeout 900 ./dropin_test 2>&1 | grep -i "two threads\|by-products:\|FAIL\|passed\|failed" | head -12) | tee $OUT/dropin_lines.txt
echo "== irregular variants"
for V in "" "0=1" "14=4" "14=5" "14=4,0=1" "14=5,0=1"; do TUNE=$V timeout 300 python tools/irregular_pmc.py 2>&1 | tail -1 | sed "s/^/[$V] /"; done | tee $OUT/irr_variants.txt
echo "== gather policy on the headline matrix"
timeout 600 python - <<'PY' | tee $OUT/gather_policy_256.txt
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
a = g.stencil_csr(ex, 3, 256)
for lay in (-1, 4, 5, -1):
    g._lib.lib().gkoc_tune_set(C.c_int(14), C.c_int64(lay))
    print(f"27pt 256^3 gather policy {lay}: {t(a):.1f} us")
g._lib.lib().gkoc_tune_set(C.c_int(14), C.c_int64(0))
PY
echo "== round 5 additions, timed"
(cd oracle/_ref/dropin && LD_LIBRARY_PATH=.:../lib:$GRAFT_REPO_ROOT/ginkgo_amd/lib timeout 900 ./round5_bench 256 30 2>&1 | tail -14) | tee $OUT/round5_additions.txt
