"""Block-Jacobi(8) on 27-pt grid^3: plain apply against the fused apply + <r,z> (timed with its
two fold launches), alone and interleaved with SpMV launches the way a CG iteration does; z in
the class of the matrix values, r in the vectors' (as solver.Cg places them).  A variant with
2 / 4 / 8 batches of groups per wave (fewer, later partial-sum writes) was measured with this
tool and rejected: profiles/r02_experiments/jacobi_apply_dot_batches_per_wave.txt.
(development / measurement tool)
  python tools/jacobi_dot_bench.py [grid=256]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import ginkgo_amd as g
from ginkgo_amd import _lib
from ginkgo_amd.executor import MEM_VALUES

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ex = g.Cdna4Executor.create(0)
a = g.stencil_csr(ex, 3, grid)
n = grid ** 3
m = g.Jacobi.build().with_max_block_size(8).on(ex).generate(a)
r = g.Dense.from_numpy(ex, np.random.default_rng(1).uniform(-1, 1, n))
z = g.Dense(ex, ex.alloc((n, 1), torch.float64, MEM_VALUES))
q = g.Dense.create(ex, (n, 1))
rho = g.Dense.create(ex, (1, 1))
nbytes = _lib.lib().gkoc_x_workspace_bytes(C.c_int64(n), C.c_size_t(8))
work = ex.alloc(((nbytes + 7) // 8,), torch.float64)
algo = 64 * n + 4 * (n // 8 + 1) + 16 * n


def timeit(name, fn, reps=40, with_spmv=False):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if with_spmv:
            a.apply(r, q)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    us = tot / reps * 1e3
    print(f"{name:52s} {us:7.1f} us  {algo / us / 1e3:7.1f} GB/s ({100 * algo / us / 1e3 / 8000:5.1f} % of 8 TB/s)", flush=True)


for spmv in (False, True):
    tag = " (after an SpMV)" if spmv else ""
    timeit("plain apply" + tag, lambda: m.apply(r, z), with_spmv=spmv)
    timeit("fused apply + dot (+ 2 fold launches)" + tag, lambda: m.apply_dot(r, z, rho, work),
           with_spmv=spmv)
zz = z.to_numpy().copy()
m.apply(r, z)
assert np.array_equal(zz, z.to_numpy()), "fused z differs from the plain apply"
print("z bit-identical to the plain apply")
