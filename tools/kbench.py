"""Per-kernel microbenchmark of the CG building blocks on the 27-pt grid^3 matrix
(HIP events on torch's current stream, which is the stream every gkoc_* call is
enqueued on).  usage: python tools/kbench.py [grid] [reps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import ginkgo_amd as g

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ex = g.Cdna4Executor.create(0)
a = g.stencil_csr(ex, 3, grid)
n = grid ** 3
nnz = a.get_num_stored_elements()
rng = np.random.default_rng(1)
vec = lambda: g.Dense.from_numpy(ex, rng.uniform(-1, 1, n))
b, x = vec(), vec()

def timeit(name, fn, nbytes):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:28s} {ms*1e3:9.1f} us  {nbytes/ms/1e6:8.1f} GB/s  ({100*nbytes/ms/1e6/8000:5.1f} % of 8 TB/s)", flush=True)


SPMV_B = 12 * nnz + 4 * (n + 1) + 16 * n
timeit("csr spmv (fresh)", lambda: a.apply(b, x), SPMV_B)
timeit("csr spmv (again)", lambda: a.apply(b, x), SPMV_B)
p, q, r, z = (vec() for _ in range(4))
timeit("csr spmv (after 4 vecs)", lambda: a.apply(b, x), SPMV_B)
timeit("csr spmv p->q", lambda: a.apply(p, q), SPMV_B)
timeit("csr spmv r->z", lambda: a.apply(r, z), SPMV_B)
timeit("csr spmv b->z", lambda: a.apply(b, z), SPMV_B)
jac = g.Jacobi.build().with_max_block_size(8).on(ex).generate(a)
sj = g.Jacobi.build().with_max_block_size(1).on(ex).generate(a)
one = g.scalar(ex, 1.0)
half = g.scalar(ex, 0.5)
rho = g.scalar(ex, 0.7)
prev = g.scalar(ex, 0.9)
res = g.Dense.create(ex, (1, 1))
stop = torch.zeros(1, dtype=torch.uint8, device=ex.device)


from ginkgo_amd._lib import call  # noqa: E402
import ctypes as C  # noqa: E402

timeit("csr spmv (after jacobi)", lambda: a.apply(b, x), SPMV_B)
print("ptrs", hex(a.values.data_ptr()), hex(a.col_idxs.data_ptr()), hex(a.row_ptrs.data_ptr()), hex(b.values.data_ptr()), hex(x.values.data_ptr()))
timeit("csr advanced spmv", lambda: a.apply(one, b, half, x), 12 * nnz + 4 * (n + 1) + 24 * n)
timeit("block-jacobi(8) apply", lambda: jac.apply(r, z), 64 * n + 4 * (n // 8 + 1) + 16 * n)
timeit("block-jacobi(8) adv apply", lambda: jac.apply(one, r, half, z), 64 * n + 4 * (n // 8 + 1) + 24 * n)
timeit("scalar jacobi apply", lambda: sj.apply(r, z), 24 * n)
timeit("dot", lambda: r.compute_dot(z, res), 16 * n)
timeit("norm2", lambda: r.compute_norm2(res), 8 * n)
timeit("add_scaled", lambda: x.add_scaled(half, p), 24 * n)
timeit("copy", lambda: x.copy_from(p), 16 * n)
timeit("fill", lambda: x.fill(0.0), 8 * n)
S = lambda: ex.stream
timeit("cg step_1", lambda: call("gkoc_cg_step_1_f64", S(), n, 1, p.values, 1, z.values, 1,
                                 rho.values, prev.values, stop), 24 * n)
timeit("cg step_2", lambda: call("gkoc_cg_step_2_f64", S(), n, 1, x.values, 1, r.values, 1,
                                 p.values, 1, q.values, 1, prev.values, rho.values, stop), 48 * n)

# ---- fused extensions, A/B against the pairs they replace
import ctypes as C  # noqa: E402
from ginkgo_amd._lib import lib  # noqa: E402
nbytes = lib().gkoc_x_workspace_bytes(C.c_int64(n), C.c_size_t(8))
work = ex.alloc(((nbytes + 7) // 8,), torch.float64)
wb = C.c_size_t(nbytes)


def pair_spmv():
    a.apply(p, q)
    p.compute_dot(q, res)


def pair_jac():
    jac.apply(r, z)
    r.compute_dot(z, res)


def pair_step2():
    call("gkoc_cg_step_2_f64", S(), n, 1, x.values, 1, r.values, 1, p.values, 1, q.values, 1,
         prev.values, rho.values, stop)
    r.compute_norm2(res)


timeit("spmv + dot (2+1 launches)", pair_spmv, 12 * nnz + 4 * (n + 1) + 16 * n)
timeit("x_csr_spmv_dot", lambda: a.apply_dot(p, q, res, work), 12 * nnz + 4 * (n + 1) + 16 * n)
timeit("jacobi + dot", pair_jac, 64 * n + 4 * (n // 8 + 1) + 16 * n)
timeit("x_jacobi_simple_apply_dot", lambda: jac.apply_dot(r, z, res, work), 64 * n + 4 * (n // 8 + 1) + 16 * n)
timeit("step_2 + norm2", pair_step2, 48 * n)
timeit("x_cg_step_2_norm", lambda: call("gkoc_x_cg_step_2_norm_f64", S(), n, x.values, r.values, p.values,
                                        q.values, prev.values, rho.values, stop, res.values, C.c_int(1),
                                        work, wb), 48 * n)

# ---- GMRES(30) + block-Jacobi(8): time per iteration (fixed iteration count)
import time  # noqa: E402
for ortho in ("mgs", "cgs"):
    gm = (g.Gmres.build().with_krylov_dim(30).with_ortho_method(ortho)
          .with_criteria(g.stop.Iteration.build().with_max_iters(60),
                         g.stop.ResidualNorm.build().with_reduction_factor(1e-30))
          .with_generated_preconditioner(jac).on(ex).generate(a))
    rhs = g.Dense.from_numpy(ex, np.ones(n))
    sol = g.Dense.from_numpy(ex, np.zeros(n))
    gm.apply(rhs, sol)
    torch.cuda.synchronize()
    sol.fill(0.0)
    t0 = time.perf_counter()
    gm.apply(rhs, sol)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"GMRES(30) {ortho:4s} + block-Jacobi(8): {gm.num_iterations} its, {dt*1e6/gm.num_iterations:9.1f} us/it")
