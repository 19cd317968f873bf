"""f64 / f32 / mixed CSR SpMV on the 27-pt 256^3 matrix: the load layout of round 3 (GKOC_TUNE_CSR_LOAD_GROUPS
= 0) against the one of rounds 1-2 (= 1), interleaved repetitions in one process.  (The other variants
listed in profiles/r03_experiments.txt were instantiated through the same key while they were measured.)
Development tool."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import ginkgo_amd as g

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ex = g.Cdna4Executor.create(0)
n = grid ** 3
a = g.stencil_csr(ex, 3, grid, dtype=torch.float32, index_dtype=torch.int32)
nnz = a.get_num_stored_elements()
xs = np.random.default_rng(1).uniform(-1, 1, n)


def run(op, x, y, tag, nbytes, ref=None):
    for v in (0, 0, 1, 0, 1, 0, 1):
        assert g._lib.lib().gkoc_tune_set(C.c_int(2), C.c_int64(v)) == 0
        for _ in range(10):
            op.apply(x, y)
        torch.cuda.synchronize()
        if ref is None:
            ref = y.values.clone()
        same = bool(torch.equal(ref, y.values))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            op.apply(x, y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{tag} variant {v}: {ms*1e3:8.1f} us  {nbytes/ms/1e6:8.1f} GB/s ({100*nbytes/ms/1e6/8000:5.1f} %)  same bits: {same}",
              flush=True)
    g._lib.lib().gkoc_tune_set(C.c_int(2), C.c_int64(0))


a64 = g.stencil_csr(ex, 3, grid)
x64 = g.Dense.from_numpy(ex, xs)
y64 = g.Dense.create(ex, (n, 1))
run(a64, x64, y64, "f64 x f64", nnz * 12 + (n + 1) * 4 + 2 * n * 8)
del a64
x32 = g.Dense.from_numpy(ex, xs.astype(np.float32))
y32 = g.Dense.create(ex, (n, 1), torch.float32)
run(a, x32, y32, "f32 x f32", nnz * 8 + (n + 1) * 4 + 2 * n * 4)
x64 = g.Dense.from_numpy(ex, xs)
y64 = g.Dense.create(ex, (n, 1))
run(a, x64, y64, "f32 x f64", nnz * 8 + (n + 1) * 4 + 2 * n * 8)
