"""Load layouts of the single-column row-segment kernel (GKOC_TUNE_CSR_LOAD_GROUPS: 0 default, 1 = wide loads, 3 = one
entry per lane and load x 8 groups, 4 = two entries x 4 groups) for double / float values on the 27-pt 256^3 matrix
and on the Flan-like matrix (development tool)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import scipy.sparse as sp
import torch

import ginkgo_amd as g

ex = g.Cdna4Executor.create(0)
L = g._lib.lib()


def t(a, x, y, reps=10):
    for _ in range(25):
        a.apply(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        a.apply(x, y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def kron_case(dtype, grid, dof):
    l27 = g.stencil_csr(ex, 3, grid)
    l = sp.csr_matrix((l27.values.cpu().numpy(), l27.col_idxs.cpu().numpy(), l27.row_ptrs.cpu().numpy()), shape=(grid ** 3, grid ** 3))
    rng = np.random.default_rng(5)
    B = rng.uniform(0.1, 1.0, (dof, dof))
    B = B + B.T + dof * np.eye(dof)
    a = sp.kron(l, sp.csr_matrix(B), format="csr")
    a.sort_indices()
    return g.Csr.from_scipy(ex, a.astype(np.float64 if dtype == torch.float64 else np.float32))


LAYS = tuple(int(v) for v in os.environ.get('LAYS', '0,1,3,4').split(','))
cases = []
for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
    cases += [(f"7pt 200^3 {tag}", lambda dt=dt: g.stencil_csr(ex, 3, 200, restricted=True, dtype=dt)),
              (f"27pt 256^3 {tag}", lambda dt=dt: g.stencil_csr(ex, 3, 256, dtype=dt)),
              (f"27pt x B2 100^3 {tag}", lambda dt=dt: kron_case(dt, 100, 2)),
              (f"27pt x B3 80^3 {tag}", lambda dt=dt: kron_case(dt, 80, 3)),
              (f"27pt x B5 60^3 {tag}", lambda dt=dt: kron_case(dt, 60, 5))]
for name, mk in cases:
    a = mk()
    n = a.size[0]
    f32 = a.values.dtype == torch.float32
    x = g.Dense.from_numpy(ex, np.random.default_rng(1).uniform(-1, 1, n).astype(np.float32 if f32 else np.float64))
    y = g.Dense.create(ex, (n, 1), a.values.dtype)
    a.apply(x, y)
    ref = y.to_numpy().tobytes()
    best = {}
    for rep in range(3):
        for lay in LAYS:
            L.gkoc_tune_set(C.c_int(2), C.c_int64(lay))
            us = t(a, x, y)
            assert y.to_numpy().tobytes() == ref
            best.setdefault(lay, []).append(us)
    L.gkoc_tune_set(C.c_int(2), C.c_int64(0))
    nnz = a.get_num_stored_elements()
    print(f"{name:22s} n={n:9d} {nnz / n:6.1f}/row | " + " | ".join(
        f"layout {lay}: " + " ".join(f"{u:.1f}" for u in v) for lay, v in best.items()), flush=True)
    del a, x, y
