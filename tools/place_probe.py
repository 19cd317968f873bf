"""Why does the same SpMV take 0.99 ms with one output vector and 1.18 ms with
another?  Probe: same matrix, many (b, y) buffers allocated different ways."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import ginkgo_amd as g
from ginkgo_amd._lib import call, lib

grid = 256
reps = 10
n = grid ** 3
ex = g.Cdna4Executor.create(0)
dev = ex.device
rng = np.random.default_rng(1)
hb = rng.uniform(-1, 1, n)

pre_y = torch.empty(n, dtype=torch.float64, device=dev)      # before the matrix
pre_b = torch.from_numpy(hb).to(dev)
a = g.stencil_csr(ex, 3, grid)
nnz = a.get_num_stored_elements()
BYTES = 12 * nnz + 4 * (n + 1) + 16 * n


def t_spmv(bt, yt):
    args = (ex.stream, n, n, a.row_ptrs, a.col_idxs, a.values, bt, 1, yt, 1, 1)
    for _ in range(2):
        call("gkoc_csr_spmv_f64_i32", *args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call("gkoc_csr_spmv_f64_i32", *args)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def show(tag, bt, yt):
    ms = t_spmv(bt, yt)
    pb = bt.data_ptr() if hasattr(bt, "data_ptr") else bt.value
    py = yt.data_ptr() if hasattr(yt, "data_ptr") else yt.value
    print(f"{tag:34s} b={pb:#x} y={py:#x}  {ms*1e3:8.1f} us ({BYTES/ms/1e6/80:5.1f} %)", flush=True)


print(f"vals={a.values.data_ptr():#x} cols={a.col_idxs.data_ptr():#x} rp={a.row_ptrs.data_ptr():#x}")
print(torch.cuda.memory_summary(abbreviated=True)[:0])
b0 = torch.from_numpy(hb).to(dev)
res = g.Dense.create(ex, (1, 1))


def t_op(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


ys = []
for i in range(40):
    y = torch.empty(n, dtype=torch.float64, device=dev)
    ys.append(y)
    d = g.Dense(ex, y.view(n, 1))
    ms = t_spmv(b0, y)
    tf = t_op(lambda: d.fill(1.0))
    tn = t_op(lambda: d.compute_norm2(res))
    print(f"y[{i:2d}] {y.data_ptr():#x}  spmv {ms*1e3:8.1f} us ({BYTES/ms/1e6/80:5.1f} %)  fill {tf:6.1f} us  norm2 {tn:6.1f} us", flush=True)
print("--- as INPUT vector b (y = ys[39])")
for i in (0, 1, 2, 3, 4, 5, 6, 20, 39):
    ms = t_spmv(ys[i], ys[38])
    print(f"b=y[{i:2d}] -> y[38]: {ms*1e3:8.1f} us")
