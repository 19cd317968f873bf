"""Find a slow and a fast output buffer for the SpMV, then run 5 launches on
each (slow first) so a rocprofv3 --pmc pass can compare their counters.
Without arguments: just classify and print.  (development tool)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import ginkgo_amd as g
from ginkgo_amd._lib import call

grid, n = 256, 256 ** 3
ex = g.Cdna4Executor.create(0)
dev = ex.device
hb = np.random.default_rng(1).uniform(-1, 1, n)
pre = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(2)]
a = g.stencil_csr(ex, 3, grid)
b0 = torch.from_numpy(hb).to(dev)


def spmv(y):
    call("gkoc_csr_spmv_f64_i32", ex.stream, n, n, a.row_ptrs, a.col_idxs, a.values, b0, 1, y, 1, 1)


def t_spmv(y, reps=3):
    spmv(y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        spmv(y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ys = pre + [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(30)]
ts = [t_spmv(y) for y in ys]
order = np.argsort(ts)
slow, fast = ys[order[-1]], ys[order[0]]
print("times us:", " ".join(f"{t*1e3:.0f}" for t in ts))
print(f"slow y {slow.data_ptr():#x} {ts[order[-1]]*1e3:.0f} us; fast y {fast.data_ptr():#x} {ts[order[0]]*1e3:.0f} us")
torch.cuda.synchronize()
for _ in range(5):
    spmv(slow)
torch.cuda.synchronize()
for _ in range(5):
    spmv(fast)
torch.cuda.synchronize()
