"""Does it matter WHICH memory class holds which array of the 256^3 CSR SpMV, or only that they are apart?
(gko::matrix::Csr::apply through the drop-in is 3-5 % slower than the native
line in the same driver run; the drop-in's arrays sit in values 2 / col_idxs 1 / row_ptrs 0 / b 0 / x 0,
the native ones in values 0 / col_idxs 1 / row_ptrs 1 / x 2 / y 2.)  One fresh process per placement:
    python tools/class_perm.py <values role> <col role> <row_ptrs role> <x role> <y role>   (roles 1 2 3 =
    the arena's classes 0 1 2)"""
import ctypes as C
import json
import sys

sys.path.insert(0, ".")


def main():
    import numpy as np
    import torch
    import ginkgo_amd as g
    from ginkgo_amd._lib import call
    from ginkgo_amd.matrix import Csr
    rv, rc, rr, rx, ry = (int(v) for v in sys.argv[1:6])
    grid = 256
    ex = g.Cdna4Executor.create(0)
    n = grid ** 3
    # the survey must see three classes before the first placement: ask for one array of each role
    warm = [ex.alloc((1 << 18,), torch.float64, r) for r in (1, 2, 3)]
    row_ptrs = ex.alloc((n + 1,), torch.int32, rr)
    nnz = C.c_int64(0)
    call("gkoc_stencil_row_ptrs_i32", ex.stream, C.c_int(3), grid, C.c_int(0), 0, grid, row_ptrs, C.byref(nnz))
    cols = ex.alloc((nnz.value,), torch.int32, rc)
    vals = ex.alloc((nnz.value,), torch.float64, rv)
    call("gkoc_stencil_fill_f64_i32", ex.stream, C.c_int(3), grid, C.c_int(0), 0, grid, row_ptrs, cols, vals)
    a = Csr(ex, (n, n), vals, cols, row_ptrs)
    xt = ex.alloc((n, 1), torch.float64, rx)
    xt.copy_(torch.from_numpy(np.random.default_rng(42).uniform(-1, 1, n)).view(-1, 1))
    x = g.Dense(ex, xt)
    y = g.Dense(ex, ex.alloc((n, 1), torch.float64, ry))
    for _ in range(25):
        a.apply(x, y)
    best = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            a.apply(x, y)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 20)
    print(json.dumps({"roles": [rv, rc, rr, rx, ry],
                      "class_of": {k: ex.memory_class(t) for k, t in
                                   (("values", vals), ("col_idxs", cols), ("row_ptrs", row_ptrs), ("x", xt), ("y", y.values))},
                      "ms": [round(b, 4) for b in best]}))


if __name__ == "__main__":
    main()
