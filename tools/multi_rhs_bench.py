"""CSR / ELL / SELL-P SpMV on the 27-pt 256^3 Laplacian with 1, 2, 3, 4, 8 right-hand sides
(fp64 / int32, HIP events, 10 launches): the one-pass kernels (csrc/csr_spmv_multi.hpp,
fmt_spmv_multi_kernel in csrc/formats.hip); every column is checked bitwise against the
single-column product.
  python tools/multi_rhs_bench.py [grid=256]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import numpy as np
import torch

import ginkgo_amd as g

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
# further arguments: key=value settings of gkoc_tune_set to run the table with, e.g. 6=8192 2=1 (several
# keys in one setting: 6=8192,2=1)
chunks = sys.argv[2:] or [None]
ex = g.Cdna4Executor.create(0)
a = g.stencil_csr(ex, 3, grid)
n = a.size[0]
xs = np.random.default_rng(1).uniform(-1, 1, (n, 8))
single = g.Dense.create(ex, (n, 8))
for j in range(8):
    a.apply(g.Dense.from_numpy(ex, xs[:, j].copy()), single.create_submatrix((0, n), (j, j + 1)))
print(f"27-pt {grid}^3, n = {n}, nnz = {a.get_num_stored_elements()}")
formats = os.environ.get("FORMATS", "csr,ell,sellp").split(",")
for name, make in (("csr", lambda: a), ("ell", a.convert_to_ell), ("sellp", a.convert_to_sellp)):
  if name not in formats:
      continue
  op = make()
  for chunk in chunks:
    if chunk is not None:
        for kv in chunk.split(","):
            key, val = kv.split("=")
            assert g._lib.lib().gkoc_tune_set(C.c_int(int(key)), C.c_int64(int(val))) == 0
        print(f" tuning {chunk}")
    for k in ((1, 2, 3, 4, 8) if chunk in (None, chunks[0]) else (2, 3, 4, 8)):
        x = g.Dense.from_numpy(ex, xs[:, :k].copy())
        y = g.Dense.create(ex, (n, k))
        for _ in range(3):
            op.apply(x, y)
        torch.cuda.synchronize()
        same = bool(torch.equal(y.values, single.values[:, :k]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            op.apply(x, y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"  {name:6s} nrhs {k}: {ms:7.3f} ms  ({ms / k:6.3f} ms per column)  columns == single-column bits: {same}",
              flush=True)
  del op
  torch.cuda.empty_cache()
