"""Ginkgo's own distributed classes on this backend, launched with mpiexec (row a15 / (e)):
gko::experimental::distributed::{Matrix, Vector}, distributed Cg + Schwarz(block-Jacobi) and
Ginkgo's examples/distributed-solver, from the UNMODIFIED core built with GINKGO_BUILD_MPI=1
(oracle/build_ref_mpi.py) on top of the drop-in libginkgo_hip.so.  All ranks share GPU 0; the
image's MPICH is not GPU-aware, so Ginkgo stages the halo through the host by itself
(mpi::requires_host_buffer) - what runs on the device are this backend's kernels:
local + non-local SpMV, row_gather, the vector reductions, the Krylov steps, block-Jacobi."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "mpi", "bin")
MPIEXEC = os.environ.get("MPIEXEC", "/opt/conda/bin/mpiexec")


def _need():
    if not os.path.exists(os.path.join(BIN, "mpi_dist_test")) or not os.path.exists(MPIEXEC):
        pytest.skip("oracle/build_ref_mpi.py + build_mpi_dropin.py have not been run, or no mpiexec")


@pytest.mark.parametrize("ranks,grid", [(2, 24), (3, 20), (4, 16)])
def test_distributed_matrix_vector_cg_vs_reference(ranks, grid):
    _need()
    p = subprocess.run([MPIEXEC, "-n", str(ranks), "./mpi_dist_test", str(grid)], cwd=BIN,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "ALL PASSED" in p.stdout and "FAILED" not in p.stdout, p.stdout
    assert p.stdout.count("PASSED") >= 12, p.stdout
    # the apply is bit-identical to the ReferenceExecutor's (same local / non-local split)
    m = re.search(r"distributed::Matrix::apply, 2 right-hand sides.*\(([\d.e+-]+)\)", p.stdout)
    assert m and float(m.group(1)) == 0.0, p.stdout


BIN_GA = os.path.join(ROOT, "oracle", "_ref", "mpi_ga", "bin")


def _routes(out):
    m = re.search(r"gkoc_mpi routes: all-reduce rccl (\d+) staged (\d+), all-to-all-v rccl (\d+) staged (\d+), "
                  r"other staged (\d+), bytes through the host (\d+), host calls (\d+)", out)
    assert m, out[-2000:]
    return dict(zip(("ar_rccl", "ar_staged", "a2a_rccl", "a2a_staged", "other", "host_bytes", "host_calls"),
                    map(int, m.groups())))


def _need_ga():
    if not os.path.exists(os.path.join(BIN_GA, "mpi_dist_test")) or not os.path.exists(MPIEXEC):
        pytest.skip("the GPU-aware flavor (oracle/_ref/mpi_ga) has not been built, or no mpiexec")


@pytest.mark.parametrize("ranks,grid", [(2, 24), (3, 20)])
def test_gpu_aware_core_hands_device_pointers_to_the_mpi_layer(ranks, grid):
    """The core built with GINKGO_HAVE_GPU_AWARE_MPI 1 (Ginkgo's GINKGO_FORCE_GPU_AWARE_MPI): no host
    staging inside Ginkgo, device pointers reach MPI_Allreduce / MPI_Ialltoallv, where
    libgkoc_mpi_rccl.so (ginkgo_amd/gko_binding/mpi_rccl.cpp) takes them.  The ranks share GPU 0
    here (RCCL refuses two ranks on a device): since round 5 the layer's device route runs on the
    library's mailbox transport (csrc/comm_ipc.hpp) - every all-reduce of the distributed Vector and
    every all-to-all-v of the RowGatherer stays on the device, NOT A BYTE through the host;
    GKOC_MPI_TRANSPORT=rccl keeps to RCCL, i.e. here to the layer's staging through pinned memory.
    Same results either way, read_distributed on the device included."""
    _need_ga()
    for transport in ("", "rccl"):
        env = dict(os.environ, GKOC_IPC_PATIENCE_MS="60000", HSA_ENABLE_IPC_MODE_LEGACY="0")
        if transport:
            env["GKOC_MPI_TRANSPORT"] = transport
        p = subprocess.run([MPIEXEC, "-n", str(ranks), "./mpi_dist_test", str(grid)], cwd=BIN_GA,
                           capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
        assert "GPU-aware MPI: 1" in p.stdout
        assert "ALL PASSED" in p.stdout and "FAILED" not in p.stdout, p.stdout
        r = _routes(p.stdout)
        if transport == "rccl":
            assert r["ar_staged"] > 0 and r["a2a_staged"] > 0 and r["ar_rccl"] == 0 and r["a2a_rccl"] == 0, r
        else:
            assert r["ar_rccl"] > 0 and r["a2a_rccl"] > 0 and r["ar_staged"] == 0 and r["a2a_staged"] == 0, r
            assert r["host_bytes"] == 0, r
        m = re.search(r"distributed::Matrix::apply, 2 right-hand sides.*\(([\d.e+-]+)\)", p.stdout)
        assert m and float(m.group(1)) == 0.0, p.stdout


def test_mpi_layer_routes_device_buffers_over_rccl():
    """One rank with GKOC_MPI_MODE=rccl: every device buffer of Ginkgo's own MPI calls (the
    all-reduces of the distributed Vector, the all-to-all-v of the RowGatherer) goes through a real
    RCCL communicator of the layer and NOT A BYTE through the host - what each rank of a
    one-GPU-per-rank run does, minus the peers."""
    _need_ga()
    env = dict(os.environ, GKOC_MPI_MODE="rccl")
    p = subprocess.run([MPIEXEC, "-n", "1", "./mpi_dist_test", "20"], cwd=BIN_GA, env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "ALL PASSED" in p.stdout and "FAILED" not in p.stdout, p.stdout
    r = _routes(p.stdout)
    assert r["ar_rccl"] > 0 and r["ar_staged"] == 0 and r["a2a_staged"] == 0 and r["host_bytes"] == 0, r


def test_ginkgos_distributed_solver_example_gpu_aware():
    """Ginkgo's examples/distributed-solver, unmodified, on the GPU-aware core + the MPI layer"""
    _need_ga()
    p = subprocess.run([MPIEXEC, "-n", "2", "./distributed-solver", "hip", "2000"], cwd=BIN_GA,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    its = int(re.search(r"Iteration count: (\d+)", p.stdout).group(1))
    res = float(re.search(r"Final Res norm: ([\d.e+-]+)", p.stdout).group(1))
    assert res < 1e-6 and 100 < its < 200, (its, res)


def test_ginkgos_distributed_solver_example():
    """examples/distributed-solver/distributed-solver.cpp, unmodified: same iteration count on
    `hip` (this backend) and `reference`"""
    _need()
    its = {}
    for ex in ("hip", "reference"):
        p = subprocess.run([MPIEXEC, "-n", "2", "./distributed-solver", ex, "2000"], cwd=BIN,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        its[ex] = int(re.search(r"Iteration count: (\d+)", p.stdout).group(1))
        res = float(re.search(r"Final Res norm: ([\d.e+-]+)", p.stdout).group(1))
        assert res < 1e-6
    assert abs(its["hip"] - its["reference"]) <= 1, its


@pytest.mark.parametrize("ranks", [2, 3, 5])
def test_mpi_layer_by_itself_on_the_mailbox_transport(ranks):
    """tests/dropin/mpi_layer_test.cpp: MPI_Allreduce / Alltoall / (I)alltoallv / Ineighbor_alltoallv with DEVICE
    buffers against what MPI defines - ranks sharing the GPU talk through the library's mailboxes, so the device
    route of the layer runs here: the nonblocking agreement that travels in the request, two requests completed
    in different orders on different ranks, MPI_Test before completion, a datatype freed before the wait"""
    _need_ga()
    exe = os.path.join(BIN_GA, "mpi_layer_test")
    if not os.path.exists(exe):
        pytest.skip("mpi_layer_test has not been built")
    env = dict(os.environ, GKOC_MPI_VERBOSE="1", GKOC_IPC_PATIENCE_MS="60000", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([MPIEXEC, "-n", str(ranks), exe], cwd=BIN_GA, capture_output=True, text=True, timeout=600,
                       env=env)
    assert p.returncode == 0 and "MPI LAYER: ALL PASSED" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
    assert "mailboxes in peer-mapped device memory" in p.stderr
    assert "staged 0 | all-to-all-v rccl" in p.stderr           # nothing went through the host
    assert "stopped waiting for a peer" not in p.stderr        # no kernel of the transport ran out of patience
