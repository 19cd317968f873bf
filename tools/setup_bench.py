"""Set-up kernels timed on the 27-pt grid^3 matrix
(development tool)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import ginkgo_amd as g

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ex = g.Cdna4Executor.create(0)


def tm(name, fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    print(f"{name:44s} {(time.perf_counter()-t)*1e3/reps:9.2f} ms", flush=True)
    return r


a = tm("stencil generator (row_ptrs + fill)", lambda: g.stencil_csr(ex, 3, grid))
n = grid ** 3
tm("csr::is_sorted_by_column_index", lambda: a.is_sorted_by_column_index())
tm("csr::extract_diagonal", lambda: a.extract_diagonal())
tm("csr::convert_to_ell", lambda: a.convert_to_ell(), 2)
torch.cuda.empty_cache()
tm("csr::convert_to_sellp (+ slice sets)", lambda: a.convert_to_sellp(), 2)
torch.cuda.empty_cache()
tm("jacobi block(8): find_blocks + generate", lambda: g.Jacobi.build().with_max_block_size(8).with_skip_sorting(True).on(ex).generate(a), 2)
tm("jacobi scalar: extract + invert", lambda: g.Jacobi.build().with_max_block_size(1).with_skip_sorting(True).on(ex).generate(a), 2)
b = g.Csr(ex, a.size, a.values.clone(), a.col_idxs.clone(), a.row_ptrs)
tm("csr::sort_by_column_index (already sorted)", lambda: b.sort_by_column_index(), 2)
x = g.Dense.from_numpy(ex, np.ones(n))
idx = torch.arange(0, n, 7, device=ex.device, dtype=torch.int32)
out = g.Dense.create(ex, (idx.numel(), 1))
tm("dense::row_gather (every 7th row)", lambda: x.row_gather(idx, out), 10)
