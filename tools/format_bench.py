"""All sparse formats on the 27-pt grid^3 Laplacian, fp64 / int32: kernel time (HIP
events, 20 launches) against each format's own algorithmic bytes.
  Csr     12 nnz + 4 (n+1) + 16 n
  Coo     16 nnz + 16 n                (row index per entry; workspace traffic not counted)
  Ell     12 n k + 16 n
  Sellp   12 sum_s 64 len_s + 16 n
  Hybrid  Ell part + Coo part (+ 8 n: the Coo part re-reads c)
  python tools/format_bench.py [grid=256]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import ginkgo_amd as g

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ex = g.Cdna4Executor.create(0)
a = g.stencil_csr(ex, 3, grid)
n, nnz = grid ** 3, a.get_num_stored_elements()
x = g.Dense.from_numpy(ex, np.random.default_rng(42).uniform(-1, 1, n))
y = g.Dense.create(ex, (n, 1))
ref = g.Dense.create(ex, (n, 1))
a.apply(x, ref)
print(f"27-pt {grid}^3: n = {n}, nnz = {nnz}")


def timeit(name, op, nbytes, run=None):
    run = run or (lambda: op.apply(x, y))
    for _ in range(25):      # the first launches after an idle phase run at lower clocks
        run()
    torch.cuda.synchronize()
    same = bool(torch.equal(y.values, ref.values))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"  {name:22s} {ms:7.3f} ms  {nbytes / 1e9:6.3f} GB  {nbytes / ms / 1e6:7.1f} GB/s "
          f"({100 * nbytes / ms / 1e6 / 8000:5.1f} % of 8 TB/s)  bits == csr: {same}", flush=True)


timeit("csr", a, 12 * nnz + 4 * (n + 1) + 16 * n)
coo = a.convert_to_coo()
timeit("coo", coo, 16 * nnz + 16 * n)
one, zero = g.Dense.from_numpy(ex, np.array([1.0])), g.Dense.from_numpy(ex, np.array([0.0]))
# y = 1 A x + 0 y: the same bits, through the operation that (in general) reads y
timeit("coo advanced", coo, 16 * nnz + 16 * n, lambda: coo.apply(one, x, zero, y))
del coo
torch.cuda.empty_cache()
ell = a.convert_to_ell()
timeit("ell", ell, 12 * n * ell.num_stored_per_row + 16 * n)
del ell
torch.cuda.empty_cache()
sp = a.convert_to_sellp()
timeit("sellp", sp, 12 * int(sp.values.numel()) + 16 * n)
del sp
torch.cuda.empty_cache()
for lim in (27, 18):
    h = a.convert_to_hybrid(column_limit=lim)
    cn = h.coo.get_num_stored_elements()
    timeit(f"hybrid(ell {lim} + coo {cn})", h, 12 * n * lim + 16 * n + (16 * cn + 24 * n if cn else 0))
    del h
    torch.cuda.empty_cache()
