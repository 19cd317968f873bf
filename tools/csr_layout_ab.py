"""A/B of the CSR kernel's load layout (GKOC_TUNE_CSR_LOAD_GROUPS 0 / 1) in one process on the
27-pt stencil at several sizes and on the Flan-like matrix (development tool)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import scipy.sparse as sp
import torch

import ginkgo_amd as g

ex = g.Cdna4Executor.create(0)


def ab(name, a):
    n = a.size[0]
    nnz = a.get_num_stored_elements()
    x = g.Dense.from_numpy(ex, np.random.default_rng(1).uniform(-1, 1, n))
    y = g.Dense.create(ex, (n, 1))
    nbytes = 12 * nnz + 4 * (n + 1) + 16 * n
    out = []
    for v in (0, 1, 0, 1, 0, 1):
        assert g._lib.lib().gkoc_tune_set(C.c_int(2), C.c_int64(v)) == 0
        for _ in range(30):
            a.apply(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            a.apply(x, y)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 30 * 1e3)
    g._lib.lib().gkoc_tune_set(C.c_int(2), C.c_int64(0))
    new, old = min(out[0::2]), min(out[1::2])
    print(f"{name:28s} n = {n:9d}: 2x3 {new:8.1f} us ({100*nbytes/new/8e6:5.1f} %)   4x1 {old:8.1f} us ({100*nbytes/old/8e6:5.1f} %)   "
          f"all: {' '.join(f'{t:.1f}' for t in out)}", flush=True)


for grid in (64, 101, 128, 160, 203, 256):
    ab(f"27-pt {grid}^3", g.stencil_csr(ex, 3, grid))
grid = 80
l27 = g.stencil_csr(ex, 3, grid)
l = sp.csr_matrix((l27.values.cpu().numpy(), l27.col_idxs.cpu().numpy(), l27.row_ptrs.cpu().numpy()),
                  shape=(grid ** 3, grid ** 3))
B3 = np.array([[4.0, 1.0, 0.5], [1.0, 3.0, 0.25], [0.5, 0.25, 2.0]])
a = sp.kron(l, sp.csr_matrix(B3), format="csr")
a.sort_indices()
ab("Flan-like (81 / row)", g.Csr.from_scipy(ex, a))
ab("7-pt 256^3", g.stencil_csr(ex, 3, 256, points=7) if "points" in g.stencil_csr.__code__.co_varnames else g.stencil_csr(ex, 2, 4096))
