#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/sp.py <<'PY'
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import ginkgo_amd as g
ex = g.Cdna4Executor.create(0)
for grid, nz in ((256, 16), (256, 32), (256, 64), (256, 256)):
    a = g.stencil_csr(ex, 3, grid, z0=96, nz=nz) if nz < grid else g.stencil_csr(ex, 3, grid)
    n = a.size[0]
    # square local block: columns re-based like the distributed split does (only timing matters)
    x = g.Dense.from_numpy(ex, np.random.default_rng(1).uniform(-1, 1, a.size[1]))
    y = g.Dense.create(ex, (n, 1))
    for _ in range(5): a.apply(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): a.apply(x, y)
    e1.record(); torch.cuda.synchronize()
    nnz = a.get_num_stored_elements()
    t = e0.elapsed_time(e1) * 10
    print(f"GKOC_TUNE_2={os.environ.get('GKOC_TUNE_2','0')} rows {n:9d}: {t:8.1f} us  {(12*nnz+20*n)/t/1e6:6.2f} TB/s")
PY
for k in 0 3; do GKOC_TUNE_2=$k python /tmp/sp.py 2>&1 | grep -v amdgpu; done
exit 0
