"""Step kernels and whole iterations of Bicgstab / Cgs / Fcg / PipeCg (and Cg for
scale) on the 27-pt grid^3 Laplacian with block-Jacobi(8): per-kernel time and
fraction of 8 TB/s (algorithmic bytes = values read + written per element, header of
csrc/krylov_steps.hip), then iterations/s over a fixed iteration count.
  python tools/family_bench.py [grid=256] [iters=60]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import time

import numpy as np
import torch

import ginkgo_amd as g
from ginkgo_amd._lib import call
from krylov_family_abi import KERNELS

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ex = g.Cdna4Executor.create(0)
n = grid ** 3
a = g.stencil_csr(ex, 3, grid)
rng = np.random.default_rng(1)
print(f"grid {grid}^3, n = {n}")
vecs = {}


def vec(name):
    if name not in vecs:
        vecs[name] = g.Dense.from_numpy(ex, rng.uniform(-1, 1, n))
    return vecs[name]


for solver, kernels in KERNELS.items():
    for kernel, spec in kernels.items():
        if kernel == "finalize":
            continue            # a no-op unless a column has just stopped
        args, nvals = [], 0
        for name, kind in spec:
            if kind in "Vv":
                d = vec(name)
                args += [d.values, 1]
                nvals += 1
            elif kind in "Ss":
                args.append(g.scalar(ex, 0.7).values)
            else:
                args.append(ex.zeros((1,), torch.uint8))
        # in-place operands are read and written
        reads = sum(1 for nme, k in spec if k == "V") + \
            (0 if kernel.startswith("initialize") else sum(1 for nme, k in spec if k == "v"))
        if solver == "pipe_cg" and kernel == "step_1":
            reads -= 1          # z2 is written only
        if kernel in ("step_2",) and solver in ("bicgstab", "cgs"):
            reads = sum(1 for nme, k in spec if k == "V")      # s / q, t are outputs only
        if solver == "cgs" and kernel == "step_1":
            reads = 3           # r, q, p ; u is written only
        if solver == "fcg" and kernel == "step_2":
            reads = 4           # x, r, p, q ; t written only
        if solver == "bicgstab" and kernel == "step_3":
            reads = 5           # x, s, t, y, z ; r written only
        writes = sum(1 for nme, k in spec if k == "v")
        fn = lambda: call(f"gkoc_{solver}_{kernel}_f64", ex.stream, n, 1, *args)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        nbytes = 8 * n * (reads + writes)
        print(f"  {solver + '::' + kernel:24s} {ms * 1e3:8.1f} us  {reads}r+{writes}w values/elem "
              f"{nbytes / ms / 1e6:8.1f} GB/s ({100 * nbytes / ms / 1e6 / 8000:5.1f} % of 8 TB/s)", flush=True)

vecs.clear()
torch.cuda.empty_cache()
rhs = g.Dense.from_numpy(ex, np.ones(n))
for name, cls in (("Cg", g.Cg), ("Fcg", g.Fcg), ("PipeCg", g.PipeCg), ("Bicgstab", g.Bicgstab), ("Cgs", g.Cgs)):
    s = (cls.build().with_criteria(g.stop.Iteration.build().with_max_iters(iters),
                                   g.stop.ResidualNorm.build().with_reduction_factor(1e-30))
         .with_preconditioner(g.Jacobi.build().with_max_block_size(8)).on(ex).generate(a))
    x = g.Dense.from_numpy(ex, np.zeros(n))
    s.apply(rhs, x)                       # warm-up (workspace)
    x.fill(0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.apply(rhs, x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    spmv = 2 if name in ("Bicgstab", "Cgs") else 1
    print(f"{name:9s} {s.num_iterations:4d} iterations  {dt / s.num_iterations * 1e3:7.3f} ms/it  "
          f"{s.num_iterations / dt:7.1f} it/s   ({spmv} SpMV + {spmv} block-Jacobi per iteration)", flush=True)
    del s, x
    torch.cuda.empty_cache()
