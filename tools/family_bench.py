"""Step kernels of Bicgstab / Cgs / Fcg / PipeCg / Bicg / Gcr / Minres and whole iterations of every
solver of ginkgo_amd (Cg for scale) on the 27-pt grid^3 Laplacian with block-Jacobi(8): per-kernel time and
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
# values read + written per element (vector operands; in-place operands count twice)
TRAFFIC = {
    ("bicgstab", "initialize"): (1, 8), ("bicgstab", "step_1"): (3, 1), ("bicgstab", "step_2"): (2, 1),
    ("bicgstab", "step_3"): (5, 2),
    ("bicg", "initialize"): (1, 8), ("bicg", "step_1"): (4, 2), ("bicg", "step_2"): (6, 3),
    ("gcr", "initialize"): (1, 1), ("gcr", "restart"): (2, 2), ("gcr", "step_1"): (4, 2),
    ("minres", "initialize"): (2, 6), ("minres", "step_2"): (7, 6),
    ("cgs", "initialize"): (1, 8), ("cgs", "step_1"): (3, 2), ("cgs", "step_2"): (2, 2), ("cgs", "step_3"): (4, 2),
    ("fcg", "initialize"): (1, 5), ("fcg", "step_1"): (2, 1), ("fcg", "step_2"): (4, 3),
    ("pipe_cg", "initialize_1"): (1, 1), ("pipe_cg", "initialize_2"): (4, 4), ("pipe_cg", "step_1"): (8, 5),
    ("pipe_cg", "step_2"): (8, 4),
}


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
                args.append(ex.zeros((8,), torch.uint8))      # stop byte / uint64 counter
        if (solver, kernel) not in TRAFFIC:
            continue            # scalar-only kernels (minres::step_1)
        reads, writes = TRAFFIC[(solver, kernel)]
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
SOLVERS = (("Cg", g.Cg, {}, 1), ("Fcg", g.Fcg, {}, 1), ("PipeCg", g.PipeCg, {}, 1),
           ("Minres", g.Minres, {}, 1), ("Bicgstab", g.Bicgstab, {}, 2), ("Cgs", g.Cgs, {}, 2),
           ("Bicg", g.Bicg, {}, 2), ("Gcr(10)", g.Gcr, {"krylov_dim": 10}, 1),
           ("Ir(0.9)", g.Ir, {"relaxation_factor": 0.9}, 1), ("Chebyshev", g.Chebyshev, {"foci": (0.02, 2.0)}, 1))
for name, cls, params, spmv in SOLVERS:
    f = cls.build().with_criteria(g.stop.Iteration.build().with_max_iters(iters),
                                  g.stop.ImplicitResidualNorm.build().with_reduction_factor(1e-30)
                                  if name == "Minres" else
                                  g.stop.ResidualNorm.build().with_reduction_factor(1e-30))
    for k, val in params.items():
        f = getattr(f, "with_" + k)(val)
    s = f.with_preconditioner(g.Jacobi.build().with_max_block_size(8)).on(ex).generate(a)
    x = g.Dense.from_numpy(ex, np.zeros(n))
    s.apply(rhs, x)                       # warm-up (workspace)
    x.fill(0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.apply(rhs, x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:10s} {s.num_iterations:4d} iterations  {dt / s.num_iterations * 1e3:7.3f} ms/it  "
          f"{s.num_iterations / dt:7.1f} it/s   ({spmv} SpMV per iteration)", flush=True)
    del s, x
    torch.cuda.empty_cache()
