import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import ginkgo_amd as g
ex = g.Cdna4Executor.create(0)
for grid in (40, 64, 128):
    a = g.stencil_csr(ex, 3, grid); n = grid**3
    jac = g.Jacobi.build().with_max_block_size(8).on(ex).generate(a)
    rhs = g.Dense.from_numpy(ex, np.ones(n)); sol = g.Dense.from_numpy(ex, np.zeros(n))
    for mode in (False, True, "auto"):
        s = (g.Cg.build().with_hip_graph(mode)
             .with_criteria(g.stop.Iteration.build().with_max_iters(300),
                            g.stop.ResidualNorm.build().with_reduction_factor(1e-300))
             .with_generated_preconditioner(jac).on(ex).generate(a))
        s.apply(rhs, sol.fill(0.0)); torch.cuda.synchronize()
        t = time.perf_counter(); s.apply(rhs, sol.fill(0.0)); torch.cuda.synchronize(); t = time.perf_counter() - t
        print(f"grid {grid:4d} hip_graph={mode!s:5s}: {s.num_iterations} its {t*1e6/s.num_iterations:8.1f} us/it")
