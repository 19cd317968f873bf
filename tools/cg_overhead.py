"""Host-overhead check of the CG loops at the per-rank size of an 8-GPU run
(256^3 / 8 ranks = 2.1 M rows = 128^3).  (development tool)
usage: python tools/cg_overhead.py [grid=128] [iters=200]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import torch.distributed as dist

import ginkgo_amd as g
from ginkgo_amd import distributed as gd

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29631")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
ex = g.Cdna4Executor.create(0)
n = grid ** 3
a = g.stencil_csr(ex, 3, grid)


def sync():
    torch.cuda.synchronize()


rhs = g.Dense.from_numpy(ex, np.ones(n))
sol = g.Dense.from_numpy(ex, np.zeros(n))
jac_f = g.Jacobi.build().with_max_block_size(8).on(ex).generate(a)
for fused in (False, True, False, True):
    solver = (g.Cg.build().with_fused_kernels(fused)
              .with_criteria(g.stop.Iteration.build().with_max_iters(iters),
                             g.stop.ResidualNorm.build().with_reduction_factor(1e-30))
              .with_generated_preconditioner(jac_f)
              .on(ex).generate(a))
    solver.apply(rhs, sol.fill(0.0))
    sync()
    t = time.perf_counter()
    solver.apply(rhs, sol.fill(0.0))
    sync()
    t = time.perf_counter() - t
    print(f"plain Cg fused={fused!s:5s} grid {grid}: {solver.num_iterations} its, {t*1e6/solver.num_iterations:8.1f} us/it, {solver.num_iterations/t:8.1f} it/s")

part = gd.SlabPartition(grid, 1)
op = gd.DistributedStencil(ex, part, 0)
op.prepare_cg(iters, sync)
its, tcg = op.timed_cg(sync)
print(f"DistributedCg(1)   grid {grid}: {its} its, {tcg*1e6/its:8.1f} us/it, {its/tcg:8.1f} it/s")

# GPU-only time of one iteration's kernels (events around a replay without host syncs)
r, z, p, q, x = (g.Dense.from_numpy(ex, np.random.default_rng(i).uniform(-1, 1, n)) for i in range(5))
jac = g.Jacobi.build().with_max_block_size(8).on(ex).generate(a)
rho, beta, prev, tau = (g.scalar(ex, 0.5) for _ in range(4))
stop = torch.zeros(1, dtype=torch.uint8, device=ex.device)
from ginkgo_amd._lib import call


def one_iter():
    jac.apply(r, z)
    r.compute_dot(z, rho)
    r.compute_norm2(tau)
    call("gkoc_cg_step_1_f64", ex.stream, n, 1, p.values, 1, z.values, 1, rho.values, prev.values, stop)
    a.apply(p, q)
    p.compute_dot(q, beta)
    call("gkoc_cg_step_2_f64", ex.stream, n, 1, x.values, 1, r.values, 1, p.values, 1, q.values, 1,
         beta.values, rho.values, stop)


for _ in range(5):
    one_iter()
sync()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t = time.perf_counter()
e0.record()
for _ in range(100):
    one_iter()
e1.record()
t_issue = time.perf_counter() - t
sync()
print(f"kernels only (no host sync): GPU {e0.elapsed_time(e1)*10:8.1f} us/it; host issue {t_issue*1e4:8.1f} us/it")
dist.destroy_process_group()
