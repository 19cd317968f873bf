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
