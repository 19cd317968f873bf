import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import gko_oracle
    gko_oracle.lib()
    return gko_oracle


@pytest.fixture(scope="session")
def gexec():
    """Cdna4Executor on cuda:0; fails loudly if the HIP library is missing."""
    import ginkgo_amd as g
    return g.Cdna4Executor.create(0)


# Order of the GPU suite: the parity tests of SURVEY.md 8's rows run
# first, each file in the order of the rows it proves, the multi-process plumbing (ranks sharing one GPU,
# bench.py as a subprocess, mpiexec) last - so that under `-x` a failure in the plumbing cannot hide the
# hot path's parity evidence.  Files not listed keep their alphabetical place between the two groups.
_ORDER = [
    "test_spmv_gpu", "test_krylov_gpu", "test_gmres_gpu", "test_jacobi_mfma_gpu", "test_fullsize_gpu",
    "test_krylov_family_gpu", "test_reduce_one_kernel_gpu", "test_mixed_gpu",
    "test_conversions_gpu", "test_coo_hybrid_gpu", "test_assembly_gpu", "test_complex_gpu",
    "test_dropin_gpu", "test_reftests_gpu", "test_native_cg_gpu", "test_flan_like_gpu",
]
_LAST = [
    "test_benchmark_harness", "test_benchmark_driver_gpu", "test_mpi_dropin_gpu", "test_mpi_reftests_gpu",
    "test_arena_roles_gpu", "test_arena_classes_gpu", "test_comm_mailbox_gpu", "test_distributed",
]
# inside test_distributed.py: the tests that start bench.py itself go at the very end
_BENCH_LAST = ("test_bench_",)


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if mod in _ORDER:
            grp = (0, _ORDER.index(mod))
        elif mod in _LAST:
            grp = (2, _LAST.index(mod))
        else:
            grp = (1, 0)
        tail = 1 if (grp[0] == 2 and item.name.startswith(_BENCH_LAST)) else 0
        return (grp[0], tail if grp[0] == 2 else 0, grp[1])
    items.sort(key=key)          # stable: the order inside a file is kept
