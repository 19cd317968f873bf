"""one rank of 8 (256^3): the local block, the ext matrix ([local | halo] columns) through the plain CSR
kernel, and the gated one-kernel product with the gate opened in advance (development probe)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import ginkgo_amd as g
from ginkgo_amd import distributed as gd
from ginkgo_amd._lib import call

ex = g.Cdna4Executor.create(0)
grid, world, rank = 256, 8, 3
part = gd.SlabPartition(grid, world)
z0, z1 = part.plane_offsets[rank], part.plane_offsets[rank + 1]
lo, hi = part.range_of(rank)
owned = g.stencil_csr(ex, 3, grid, z0=z0, nz=z1 - z0)
be = gd.HipBackend(ex)
local, nl, recv_gidx = be.split(owned, lo, hi, grid ** 3)
n = hi - lo
e = nl["ext"]
ext = g.Csr(ex, (n, e["n_cols"]), e["vals"], e["cols"], e["ptrs"])
print("classes local", local.memory_classes(), "ext", ext.memory_classes())
store = ex.zeros((e["n_cols"],), torch.float64)
store.copy_(torch.rand(e["n_cols"], dtype=torch.float64, device=store.device))
x = g.Dense(ex, store[:n].view(n, 1))
xe = g.Dense(ex, store.view(-1, 1))
y = g.Dense.create(ex, (n, 1))
y2 = g.Dense.create(ex, (n, 1))
gate = be.gate_new()
side = be.side_stream()


def tm(name, fn, reps=50):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"   {name:58s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us", flush=True)


tm("local block (local columns only)", lambda: local.apply(x, y))
tm("ext matrix, plain CSR kernel over [x | halo]", lambda: ext.apply(xe, y))


def gated():
    be.gate_open(torch.cuda.current_stream(), gate)     # opened in front of the kernel, same stream
    be.spmv_gated(nl, store, y2, gate)


tm("gated one-kernel product, gate opened in advance", gated)
print("same bits as the plain kernel on the ext matrix:", bool(torch.equal(y.values, y2.values)))
