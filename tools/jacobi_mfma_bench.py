"""Block-Jacobi(8) apply on 27-pt grid^3 with 1 .. 32 right-hand sides: the lane = row kernels
(bit-identical to the reference) against the matrix-core kernel (GKOC_TUNE_JACOBI_MFMA = 1,
v_mfma_f64_16x16x4_f64).  Bytes: 64 n (blocks, once) + 16 n nrhs (b and x).
  python tools/jacobi_mfma_bench.py [grid=256]   (development / measurement tool)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import ginkgo_amd as g
from ginkgo_amd import _lib

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ex = g.Cdna4Executor.create(0)
a = g.stencil_csr(ex, 3, grid)
n = grid ** 3
m = g.Jacobi.build().with_max_block_size(8).on(ex).generate(a)
rng = np.random.default_rng(1)
print(f"27-pt {grid}^3, n = {n}, block-Jacobi(8), {m.num_blocks} blocks")
for k in (1, 2, 4, 8, 16, 32):
    if 16 * n * k > 24e9:
        break
    b = g.Dense.from_numpy(ex, rng.uniform(-1, 1, (n, k)))
    x = g.Dense.create(ex, (n, k))
    algo = 64 * n + 4 * (n // 8 + 1) + 16 * n * k
    res = {}
    for mode in (0, 1):
        _lib.call("gkoc_tune_set", C.c_int(3), C.c_int64(mode))
        for _ in range(25):      # the first launches after an idle phase run at lower clocks
            m.apply(b, x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            m.apply(b, x)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        res[mode] = (us, x.to_numpy() if n * k <= 2 ** 28 else None)
    _lib.call("gkoc_tune_set", C.c_int(3), C.c_int64(0))
    diff = ""
    if res[0][1] is not None:
        d = np.max(np.abs(res[0][1] - res[1][1])) / np.max(np.abs(res[0][1]))
        diff = f"  max rel diff {d:.1e}"
    print(f"  nrhs {k:2d}: lane = row {res[0][0]:8.1f} us ({100 * algo / res[0][0] / 1e3 / 8000:5.1f} %)   "
          f"matrix cores {res[1][0]:8.1f} us ({100 * algo / res[1][0] / 1e3 / 8000:5.1f} % of 8 TB/s){diff}", flush=True)
