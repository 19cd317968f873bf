"""Multi-rank path: world_size-2/3 gloo runs on CPU of the
partition / halo-plan / exchange / all-reduce logic (oracle-backed kernels), and
on the GPU box a 2-rank run with the HIP kernels (both ranks on cuda:0, exchange
staged through gloo).  Mirrors test/mpi/distributed/{matrix,vector}.cpp and
test/mpi/solver/solver.cpp (ranks launched on one node)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(mode, world, grid, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), mode, str(grid)]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for attempt in range(2):       # one retry: the rendezvous port can race
        cmd[cmd.index("--master-port") + 1] = str(_free_port())
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        if p.returncode == 0:
            break
    assert p.returncode == 0, p.stdout[-3000:] + "\n--- stderr ---\n" + p.stderr[-12000:]
    assert "dist_worker OK" in p.stdout


def test_partition_helpers():
    import ginkgo_amd.distributed as gd
    p = gd.Partition.build_from_global_size_uniform(3, 10)
    assert p.offsets == [0, 4, 7, 10]          # partition.hpp:262 semantics
    assert p.owner_of(np.array([0, 3, 4, 6, 7, 9])).tolist() == [0, 0, 1, 1, 2, 2]
    s = gd.Partition.build_slabs(8, 3)
    assert s.offsets == [0, 3 * 64, 6 * 64, 8 * 64]
    with pytest.raises(Exception):
        gd.Partition([0, 5, 3])


@pytest.mark.parametrize("world,grid", [(2, 8), (3, 9)])
def test_distributed_cpu_gloo(world, grid):
    _launch("cpu", world, grid)


@pytest.mark.gpu
def test_distributed_gpu_two_ranks_one_device():
    _launch("gpu", 2, 16)


@pytest.mark.gpu
def test_distributed_single_rank_matches_plain(gexec, oracle):
    """world = 1: the distributed wrapper degenerates to the plain SpMV / CG"""
    import torch.distributed as dist
    import ginkgo_amd as g
    import ginkgo_amd.distributed as gd
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}",
                                rank=0, world_size=1)
    grid = 12
    part = gd.SlabPartition(grid, 1)
    op = gd.DistributedStencil(gexec, part, 0)
    rp, ci, v = oracle.stencil_csr(3, grid)
    assert op.global_nnz == len(v) and op.matrix.n_halo == 0
    x = op.random_vector(42)
    y = op.zeros_vector()
    op.apply(x, y)
    xg = np.random.default_rng(42).uniform(-1, 1, grid ** 3)
    assert np.array_equal(y.to_numpy()[:, 0], oracle.csr_spmv(rp, ci, v, xg))
    op.prepare_cg(5, lambda: gexec.synchronize())
    iters, t = op.timed_cg(lambda: gexec.synchronize())
    assert iters == 5
    dist.destroy_process_group()
