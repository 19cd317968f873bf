"""Host (enqueue) cost of one DistributedCg iteration under RCCL, piece by piece.
One rank, backend nccl, the mirror construction of tests/rccl_mirror_worker.py on a
grid small enough that the device is never the bottleneck: what is measured is the
time the Python driver + torch.distributed + RCCL need to ENQUEUE an iteration -
the floor of the iteration time at 8 GPUs, where the device part is ~250 us.
  python tools/dist_host_cost.py [grid=16] [iters=400] [direct]"""
import os
import sys
import time

import numpy as np
import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))


def main():
    import rccl_mirror_worker as w
    import ginkgo_amd.distributed as gd
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    direct = len(sys.argv) > 3 and sys.argv[3] == "direct"
    w.init_rccl_single()
    be, comm, a, part, calls = w.mirror_problem(grid, direct)
    print("communicator:", type(comm).__name__)
    lo, hi = part.range_of(0)
    acc = {}

    def wrap(obj, name, label=None):
        f = getattr(obj, name)
        label = label or name

        def g(*args, **kw):
            t0 = time.perf_counter()
            r = f(*args, **kw)
            d = acc.setdefault(label, [0.0, 0])
            d[0] += time.perf_counter() - t0
            d[1] += 1
            return r
        setattr(obj, name, g)

    for nm in ("gather", "spmv", "rowlist_add", "cg_step_1", "jacobi_apply_dot", "cg_step_2_sqnorm",
               "check_begin", "check_done", "local_dot", "local_sqnorm"):
        if hasattr(be, nm):
            wrap(be, nm)
    wrap(comm, "all_to_all_v")
    if direct:
        wrap(comm, "exchange_begin")
        wrap(comm, "exchange_end")
    wrap(comm, "all_reduce_sum_")
    wrap(a, "apply", "matrix.apply (total)")
    for lag in (6, 0):
        solver = gd.DistributedCg(be, comm, a, iters, 1e-300, 8, check_lag=lag)
        b = be.vector_from(np.ones(hi - lo))
        x = be.vector(hi - lo)
        solver.apply(b, x)                    # warm-up
        torch.cuda.synchronize()
        acc.clear()
        x.fill(0.0)
        t0 = time.perf_counter()
        solver.apply(b, x)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        n = solver.num_iterations
        print(f"grid {grid} lag {lag}: {n} iterations, host enqueue {t_host / n * 1e6:.1f} us/it, "
              f"until device idle {t_all / n * 1e6:.1f} us/it")
        for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
            print(f"   {k:28s} {t / n * 1e6:7.1f} us/it  ({c / n:.2f} calls/it, {t / c * 1e6:.1f} us/call)")
    import torch.distributed as dist
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
