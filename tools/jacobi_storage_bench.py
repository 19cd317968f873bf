"""block-Jacobi(8) apply and CG iterations/s on the 27-pt grid^3 Laplacian with the
inverse blocks stored as double / float / half / truncated types (fixed
storage_optimization).   python tools/jacobi_storage_bench.py [grid=256]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import ginkgo_amd as g

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ex = g.Cdna4Executor.create(0)
a = g.stencil_csr(ex, 3, grid)
n = grid ** 3
b = g.Dense.from_numpy(ex, np.random.default_rng(1).uniform(-1, 1, n))
x = g.Dense.create(ex, (n, 1))
print(f"27-pt {grid}^3, block-Jacobi(8): {n // 8} blocks; apply = 20 launches (HIP events); CG = 60 fixed iterations")
CASES = (("double (0,0)", None, 8), ("float (0,1)", (0, 1), 4), ("half (0,2)", (0, 2), 2),
         ("upper 32 bits of the double (1,0)", (1, 0), 4), ("upper 16 bits of the float (1,1)", (1, 1), 2),
         ("upper 16 bits of the double (2,0)", (2, 0), 2),
         ("autodetect, accuracy 1e-1", "autodetect", None))
for name, prec, width in CASES:
    f = g.Jacobi.build().with_max_block_size(8)
    if prec == "autodetect":
        f = f.with_storage_optimization("autodetect")
    elif prec:
        f = f.with_storage_optimization(*prec)
    torch.cuda.synchronize()
    tg = time.perf_counter()
    m = f.on(ex).generate(a)
    torch.cuda.synchronize()
    tg = time.perf_counter() - tg
    if width is None:
        chosen = torch.unique(m.precisions, return_counts=True)
        width = sum({0: 8, 1: 4, 2: 2, 0x10: 4, 0x11: 2, 0x20: 2}[int(p)] * int(c)
                    for p, c in zip(*chosen)) / m.num_blocks
        name += " -> " + ", ".join(f"{int(c)} x {int(p):#04x}" for p, c in zip(*chosen))
    for _ in range(3):
        m.apply(b, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        m.apply(b, x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nbytes = n * 8 * width + 16 * n + 4 * (n // 8 + 1)
    s = (g.Cg.build().with_criteria(g.stop.Iteration.build().with_max_iters(60),
                                    g.stop.ResidualNorm.build().with_reduction_factor(1e-30))
         .with_generated_preconditioner(m).on(ex).generate(a))
    rhs, sol = g.Dense.from_numpy(ex, np.ones(n)), g.Dense.from_numpy(ex, np.zeros(n))
    s.apply(rhs, sol)
    sol.fill(0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.apply(rhs, sol)
    torch.cuda.synchronize()
    print(f"  blocks stored as {name:36s}: apply {ms * 1e3:6.1f} us, {nbytes / 1e9:5.3f} GB, "
          f"{nbytes / ms / 1e6:6.1f} GB/s; CG {60 / (time.perf_counter() - t0):6.1f} it/s; "
          f"generate (incl. find_blocks) {tg * 1e3:6.1f} ms", flush=True)
    del m, s
