import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ginkgo_amd as g
ex = g.Cdna4Executor.create(0)
grid = 256; n = grid**3
a = g.stencil_csr(ex, 3, grid)
jac = g.Jacobi.build().with_max_block_size(8).on(ex).generate(a)
for ortho in ("cgs", "mgs"):
    gm = (g.Gmres.build().with_krylov_dim(30).with_ortho_method(ortho)
          .with_criteria(g.stop.Iteration.build().with_max_iters(60),
                         g.stop.ResidualNorm.build().with_reduction_factor(1e-30))
          .with_generated_preconditioner(jac).on(ex).generate(a))
    rhs = g.Dense.from_numpy(ex, np.ones(n)); sol = g.Dense.from_numpy(ex, np.zeros(n))
    gm.apply(rhs, sol); torch.cuda.synchronize()
