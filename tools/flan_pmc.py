"""Flan-like matrix (tools/flan_bench.py), CSR and SELL-P SpMV only, a few launches each: the
command profiled by tools/r02_session6.sh under rocprofv3 --pmc (development tool)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import scipy.sparse as sp
import torch

import ginkgo_amd as g

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 80
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ex = g.Cdna4Executor.create(0)
l27 = g.stencil_csr(ex, 3, grid)
l = sp.csr_matrix((l27.values.cpu().numpy(), l27.col_idxs.cpu().numpy(), l27.row_ptrs.cpu().numpy()),
                  shape=(grid ** 3, grid ** 3))
B3 = np.array([[4.0, 1.0, 0.5], [1.0, 3.0, 0.25], [0.5, 0.25, 2.0]])
a = sp.kron(l, sp.csr_matrix(B3), format="csr")
a.sort_indices()
n = a.shape[0]
da = g.Csr.from_scipy(ex, a)
sl = da.convert_to_sellp()
x = g.Dense.from_numpy(ex, np.random.default_rng(1).uniform(-1, 1, n))
y = g.Dense.create(ex, (n, 1))
for _ in range(reps):
    da.apply(x, y)
for _ in range(reps):
    sl.apply(x, y)
torch.cuda.synchronize()
print("flan_pmc done", n, a.nnz)
