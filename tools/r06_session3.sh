#!/bin/bash
OUT=gpurun_out/r06s3
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity (spmv)"
timeout 1200 python -m pytest tests/test_spmv_gpu.py tests/test_flan_like_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $OUT/parity.txt
echo "== irregular variants"
for V in "" "0=1" "14=1" "14=2" "14=3" "14=1,0=1" "14=2,0=1" "14=1,13=2" "14=2,13=2" "14=2,13=2,0=1"; do TUNE=$V timeout 300 python tools/irregular_pmc.py 2>&1 | tail -1 | sed "s/^/[$V] /"; done | tee $OUT/irr_variants.txt
echo "== the same layouts on other short-row matrices"
timeout 600 python - <<'PY' | tee $OUT/short_rows_other.txt
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
                 ("27pt 128^3", lambda: g.stencil_csr(ex, 3, 128)), ("27pt 256^3", lambda: g.stencil_csr(ex, 3, 256))):
    a = mk()
    row = []
    for lay in (-1, 1, 2, 3):
        g._lib.lib().gkoc_tune_set(C.c_int(14), C.c_int64(lay))
        row.append(f"layout {lay}: {t(a):.1f} us")
    g._lib.lib().gkoc_tune_set(C.c_int(14), C.c_int64(0))
    print(name, " | ".join(row))
PY
