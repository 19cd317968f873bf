"""configs[4] stand-in at Flan_1565 scale: A = L27(g^3) (x) B3, g = 80:
n = 1 536 000, nnz = 123 M, up to 81 nnz/row.  CSR vs SELL-P SpMV and
CG + block-Jacobi(3) iterations/s.  (development / measurement tool)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import scipy.sparse as sp
import torch

import ginkgo_amd as g

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 80
ex = g.Cdna4Executor.create(0)
t0 = time.perf_counter()
l27 = g.stencil_csr(ex, 3, grid)
l = sp.csr_matrix((l27.values.cpu().numpy(), l27.col_idxs.cpu().numpy(), l27.row_ptrs.cpu().numpy()),
                  shape=(grid ** 3, grid ** 3))
B3 = np.array([[4.0, 1.0, 0.5], [1.0, 3.0, 0.25], [0.5, 0.25, 2.0]])
a = sp.kron(l, sp.csr_matrix(B3), format="csr")
a.sort_indices()
n, nnz = a.shape[0], a.nnz
print(f"flan-like: n={n} nnz={nnz} ({nnz/n:.1f}/row, max {np.diff(a.indptr).max()}), host build {time.perf_counter()-t0:.1f} s")
da = g.Csr.from_scipy(ex, a)
sl = da.convert_to_sellp()
x = g.Dense.from_numpy(ex, np.random.default_rng(1).uniform(-1, 1, n))
y = g.Dense.create(ex, (n, 1))


def timeit(name, fn, nbytes, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:34s} {ms*1e3:9.1f} us  {nbytes/ms/1e6:8.1f} GB/s ({100*nbytes/ms/1e6/8000:5.1f} % of 8 TB/s)")


timeit("CSR SpMV", lambda: da.apply(x, y), 12 * nnz + 4 * (n + 1) + 16 * n)
stored = sl.values.numel()
timeit(f"SELL-P SpMV (stored {stored/nnz:.3f} x nnz)", lambda: sl.apply(x, y), 12 * stored + 16 * n)
prec = g.Jacobi.build().with_max_block_size(3).on(ex).generate(da)
for name, op in (("CSR", da), ("SELL-P", sl)):
    s = (g.Cg.build()
         .with_criteria(g.stop.Iteration.build().with_max_iters(200),
                        g.stop.ResidualNorm.build().with_reduction_factor(1e-30))
         .with_generated_preconditioner(prec).on(ex).generate(op))
    rhs = g.Dense.from_numpy(ex, np.ones(n))
    sol = g.Dense.from_numpy(ex, np.zeros(n))
    s.apply(rhs, sol)
    torch.cuda.synchronize()
    sol.fill(0.0)
    t = time.perf_counter()
    s.apply(rhs, sol)
    torch.cuda.synchronize()
    t = time.perf_counter() - t
    print(f"CG + block-Jacobi(3) on {name:6s}: {s.num_iterations} its, {t*1e6/s.num_iterations:8.1f} us/it, {s.num_iterations/t:8.1f} it/s")
