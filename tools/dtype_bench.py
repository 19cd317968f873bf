"""SpMV rate for every value/index type combination of the C ABI on the 27-pt
grid^3 matrix (development tool)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import ginkgo_amd as g

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ex = g.Cdna4Executor.create(0)
n = grid ** 3
for vt, it in ((torch.float64, torch.int32), (torch.float64, torch.int64),
               (torch.float32, torch.int32), (torch.float32, torch.int64)):
    a = g.stencil_csr(ex, 3, grid, dtype=vt, index_dtype=it)
    nnz = a.get_num_stored_elements()
    vb, ib = torch.empty((), dtype=vt).element_size(), torch.empty((), dtype=it).element_size()
    x = g.Dense.from_numpy(ex, np.random.default_rng(1).uniform(-1, 1, n).astype(np.float64 if vb == 8 else np.float32))
    y = g.Dense.create(ex, (n, 1), vt)
    nbytes = nnz * (vb + ib) + (n + 1) * ib + 2 * n * vb
    for name, op in (("csr", a), ("ell", a.convert_to_ell()), ("sellp", a.convert_to_sellp())):
        for _ in range(25):      # the first launches after an idle phase run at lower clocks
            op.apply(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            op.apply(x, y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{str(vt):14s} {str(it):12s} {name:6s} {ms*1e3:8.1f} us  {nbytes/ms/1e6:8.1f} GB/s ({100*nbytes/ms/1e6/8000:5.1f} % of 8 TB/s)", flush=True)
        del op
    del a, x, y
    torch.cuda.empty_cache()

# mixed precision: float32 values, float64 vectors and arithmetic (csr / ell ::spmv<float, double,
# double>): 8 B per stored entry; compared with the all-double product of the same (widened) matrix
for it in (torch.int32,):
    a32 = g.stencil_csr(ex, 3, grid, dtype=torch.float32, index_dtype=it)
    nnz = a32.get_num_stored_elements()
    x = g.Dense.from_numpy(ex, np.random.default_rng(1).uniform(-1, 1, n))
    y = g.Dense.create(ex, (n, 1))
    nbytes = nnz * (4 + 4) + (n + 1) * 4 + 2 * n * 8
    a64 = g.stencil_csr(ex, 3, grid, index_dtype=it)   # the 27-pt entries are exact in float
    ref = g.Dense.create(ex, (n, 1))
    a64.apply(x, ref)
    del a64
    for name, op in (("csr", a32), ("ell", a32.convert_to_ell())):
        for _ in range(25):
            op.apply(x, y)
        torch.cuda.synchronize()
        same = bool(torch.equal(y.values, ref.values))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            op.apply(x, y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"f32 values x f64 vectors  {str(it):12s} {name:6s} {ms*1e3:8.1f} us  {nbytes/ms/1e6:8.1f} GB/s "
              f"({100*nbytes/ms/1e6/8000:5.1f} % of 8 TB/s)  bits == all-double product: {same}", flush=True)
