"""CSR SpMV on the 27-pt 256^3 Laplacian with 1, 2, 3, 4, 8 right-hand sides (fp64 / int32,
HIP events, 10 launches): the one-pass kernel of csrc/csr_spmv_multi.hpp against one
pass per column.
  python tools/multi_rhs_bench.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ginkgo_amd as g
ex = g.Cdna4Executor.create(0)
a = g.stencil_csr(ex, 3, 256)
n = a.size[0]
for k in (1, 2, 3, 4, 8):
    x = g.Dense.from_numpy(ex, np.random.default_rng(1).uniform(-1, 1, (n, k)))
    y = g.Dense.create(ex, (n, k))
    for _ in range(3): a.apply(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): a.apply(x, y)
    e1.record(); torch.cuda.synchronize()
    print(f"nrhs {k}: {e0.elapsed_time(e1)/10:.3f} ms", flush=True)
