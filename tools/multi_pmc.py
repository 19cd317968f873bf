"""ELL / CSR SpMV with 1, 2, 4, 8 right-hand sides on the 27-pt 256^3 Laplacian, three launches each:
the workload of tools/multi_pmc.sh (rocprofv3 --pmc passes; development tool)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import ginkgo_amd as g

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
import ctypes as C
for kv in filter(None, os.environ.get("TUNE", "").split(",")):
    key, val = kv.split("=")
    assert g._lib.lib().gkoc_tune_set(C.c_int(int(key)), C.c_int64(int(val))) == 0
ex = g.Cdna4Executor.create(0)
a = g.stencil_csr(ex, 3, grid)
n = a.size[0]
xs = np.random.default_rng(1).uniform(-1, 1, (n, 8))
ops = ((("ell", a.convert_to_ell()),) if "NOELL" not in os.environ else ()) + \
    ((("csr", a),) if "NOCSR" not in os.environ else ())
for name, op in ops:
    for k in [int(v) for v in os.environ.get("NRHS", "1,2,4,8").split(",")]:
        x = g.Dense.from_numpy(ex, xs[:, :k].copy())
        y = g.Dense.create(ex, (n, k))
        for _ in range(3):
            op.apply(x, y)
        torch.cuda.synchronize()
print("done")
