"""The heavy-tailed stand-in of configs[4] (ginkgo_amd/workloads.py irregular_rows), CSR SpMV only, a few
launches: the workload of tools/pmc_groups.sh (rocprofv3 --pmc passes / kernel trace; development tool)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import scipy.sparse as sp
import torch

import ginkgo_amd as g
from ginkgo_amd import workloads as wl

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for kv in filter(None, os.environ.get("TUNE", "").split(",")):
    key, val = kv.split("=")
    assert g._lib.lib().gkoc_tune_set(C.c_int(int(key)), C.c_int64(int(val))) == 0
ex = g.Cdna4Executor.create(0)
rp, ci, va = wl.irregular_rows(n)
a = g.Csr.from_scipy(ex, sp.csr_matrix((va, ci, rp), shape=(n, n)))
x = g.Dense.from_numpy(ex, np.random.default_rng(1).uniform(-1, 1, n))
y = g.Dense.create(ex, (n, 1))
for _ in range(3):
    a.apply(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    a.apply(x, y)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
nnz = int(rp[-1])
nbytes = 12 * nnz + 4 * (n + 1) + 16 * n
print(f"irregular n={n} nnz={nnz} csr {ms * 1e3:.1f} us = {nbytes / ms / 1e6:.0f} GB/s = {nbytes / ms / 1e6 / 80:.1f} %")
