"""one rank of 8 (256^3): the local block through the plain CSR kernel, the join-based product's two
kernels, and the gated one-kernel product with the gate opened in advance (development probe)."""
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
f = nl["full"]
print("classes local", local.memory_classes(), "gated admitted:", f.get("gated"))
store = ex.zeros((f["halo_base"] + f["n_halo"],), torch.float64)
store.copy_(torch.rand(store.numel(), dtype=torch.float64, device=store.device))
x = g.Dense(ex, store[:n].view(n, 1))
halo = g.Dense(ex, store[f["halo_base"]:].view(-1, 1))
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


def joined():
    be.spmv_rows(local, f["interior"][0], f["interior"][1], x, y)
    be.rowlist_full(nl, x, halo, y)


tm("interior rows + complete boundary rows, two kernels", joined)


def gated():
    be.gate_open(torch.cuda.current_stream(), gate)     # opened in front of the kernel, same stream
    be.spmv_gated(local, nl, store, y2, gate)


tm("gated one-kernel product, gate opened in advance", gated)
print("same bits as the two kernels:", bool(torch.equal(y.values, y2.values)))
