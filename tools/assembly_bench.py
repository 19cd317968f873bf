"""Device-side assembly of the 27-pt grid^3 Laplacian from shuffled triplets
(device_matrix_data::sort_row_major / sum_duplicates / remove_zeros, Csr::read), fp64 /
int32, timed with HIP events.  The triplets are the matrix's own entries in a random
order, every entry present twice (halves), plus one explicit zero per row - what a
finite-element assembly hands over.  Checks that the assembled CSR arrays equal the
generator's bit for bit.
  python tools/assembly_bench.py [grid=256]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import ginkgo_amd as g

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ex = g.Cdna4Executor.create(0)
a = g.stencil_csr(ex, 3, grid)
n, nnz = a.size[0], a.get_num_stored_elements()
dev = a.values.device
coo = a.convert_to_coo()
print(f"27-pt {grid}^3: n = {n}, nnz = {nnz}")


def timed(name, fn, entries):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"  {name:34s} {entries / 1e6:8.1f} M entries  {ms:8.2f} ms  {entries / ms / 1e6:6.2f} G entries/s",
          flush=True)
    return out


# 1. the matrix's own entries, shuffled: sort_row_major + Csr::read
p = torch.randperm(nnz, device=dev)
md = g.DeviceMatrixData(ex, a.size, coo.row_idxs[p].contiguous(), coo.col_idxs[p].contiguous(),
                        coo.values[p].contiguous())
del p
timed("sort_row_major (warm-up)", md.sort_row_major, nnz)
p = torch.randperm(nnz, device=dev)
md = g.DeviceMatrixData(ex, a.size, coo.row_idxs[p].contiguous(), coo.col_idxs[p].contiguous(),
                        coo.values[p].contiguous())
del p
timed("sort_row_major", md.sort_row_major, nnz)
csr = timed("Csr.read (idxs -> ptrs)", lambda: g.Csr.read(md), nnz)
ok = torch.equal(csr.row_ptrs, a.row_ptrs) and torch.equal(csr.col_idxs, a.col_idxs) and \
    torch.equal(csr.values, a.values)
print(f"  assembled CSR == generator's CSR: {ok}")
del md, csr
torch.cuda.empty_cache()

# 2. two halves per entry + an explicit zero per row: sum_duplicates + remove_zeros
rows = torch.cat((coo.row_idxs, coo.row_idxs, torch.arange(n, dtype=torch.int32, device=dev)))
cols = torch.cat((coo.col_idxs, coo.col_idxs, torch.full((n,), 7, dtype=torch.int32, device=dev)))
vals = torch.cat((coo.values * 0.25, coo.values * 0.75, torch.zeros(n, dtype=torch.float64, device=dev)))
p = torch.randperm(rows.numel(), device=dev)
md = g.DeviceMatrixData(ex, a.size, rows[p].contiguous(), cols[p].contiguous(), vals[p].contiguous())
total = rows.numel()
del rows, cols, vals, p
torch.cuda.empty_cache()
timed("sum_duplicates (sort + merge)", md.sum_duplicates, total)
timed("remove_zeros", md.remove_zeros, md.get_num_stored_elements())
csr = g.Csr.read(md)
ok = md.get_num_stored_elements() == nnz and torch.equal(csr.row_ptrs, a.row_ptrs) and \
    torch.equal(csr.col_idxs, a.col_idxs) and torch.equal(csr.values, a.values)
print(f"  assembled CSR == generator's CSR: {ok}")
