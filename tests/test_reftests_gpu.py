"""Ginkgo's OWN cross-executor tests on this backend (north_star: "drops into Ginkgo's own
examples and test harness").  The sources under /root/reference/test/** are compiled
unmodified by oracle/build_reftests.py (EXEC_TYPE=HipExecutor, GKO_DEVICE_NAMESPACE=hip, as
cmake/create_test.cmake:399-463 does) against the drop-in libginkgo_hip.so and run here.

Every suite must run to its end, and every test in it must pass unless it is listed in
tests/dropin/reftests_expected.json: those are the kernels this backend leaves to Ginkgo's
`NotCompiled` stubs (the Fbcsr format and the triangular solvers that test/matrix/matrix.cpp and
test/solver/solver.cpp instantiate next to the in-scope formats and solvers, complex IDR - outside
SURVEY.md 8).  No listed failure is
a wrong number."""
import json
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "dropin", "reftests")
EXPECTED = json.load(open(os.path.join(ROOT, "tests", "dropin", "reftests_expected.json")))


def _run(name):
    p = subprocess.run([os.path.join(BIN, name)], capture_output=True, text=True, timeout=900,
                       cwd=BIN)
    txt = p.stdout
    ran = re.search(r"^\[==========\] (\d+) tests ran", txt, re.M)
    failed = set(re.findall(r"^\[  FAILED  \] (.+)$", txt, re.M))
    failed = {f for f in failed if not re.match(r"\d+ tests, listed below:", f)}
    return p.returncode, int(ran.group(1)) if ran else None, failed, txt


@pytest.mark.parametrize("suite", sorted(EXPECTED))
def test_reference_suite(suite):
    exe = os.path.join(BIN, suite)
    if not os.path.exists(exe):
        pytest.skip("oracle/build_reftests.py has not been run (needs /root/reference)")
    rc, ran, failed, txt = _run(suite)
    exp = EXPECTED[suite]
    assert ran is not None, f"{suite} did not run to its end (rc {rc}):\n{txt[-2000:]}"
    known = set(exp["known_failures"])
    new = failed - known
    assert not new, f"{suite}: tests failing that are not known limitations: {sorted(new)}\n" + \
        txt[-3000:]
    assert ran == exp["ran"], (ran, exp["ran"])
    print(f"{suite}: {ran} ran, {ran - len(failed)} passed, {len(failed)} known NotCompiled / "
          f"NotSupported ({len(known - failed)} of the listed ones pass now)")


def test_hot_path_suites_are_fully_green():
    """the suites of the CG / GMRES / SpMV hot path have no exceptions at all"""
    for suite in ("solver_cg_kernels_hip", "solver_gmres_kernels_hip", "solver_fcg_kernels_hip",
                  "solver_pipe_cg_kernels_hip", "solver_bicgstab_kernels_hip",
                  "solver_cgs_kernels_hip", "solver_gcr_kernels_hip", "solver_ir_kernels_hip",
                  "solver_chebyshev_kernels_hip", "solver_cb_gmres_kernels_hip", "components_prefix_sum_kernels_hip",
                  "components_format_conversion_kernels_hip", "stop_criterion_kernels_hip",
                  "stop_combined_kernels_hip", "base_executor_hip", "base_timer_hip",
                  "preconditioner_jacobi_kernels_hip", "matrix_ell_kernels_hip", "matrix_sellp_kernels_hip",
                  "matrix_coo_kernels_hip", "matrix_hybrid_kernels_hip", "solver_bicg_kernels_hip",
                  "solver_minres_kernels_hip", "base_device_matrix_data_kernels_hip",
                  "components_fill_array_kernels_hip",
                  # round 3: the set-up kernels of the distributed classes, SparsityCsr, permutations
                  "distributed_assembly_kernels_hip", "distributed_index_map_kernels_hip",
                  "distributed_matrix_kernels_hip", "distributed_partition_helper_kernels_hip",
                  "distributed_partition_kernels_hip", "distributed_vector_kernels_hip",
                  "matrix_sparsity_csr_kernels_hip", "matrix_permutation_kernels_hip",
                  "matrix_scaled_permutation_kernels_hip",
                  # ... with the complex kernels and the reuse forms of SpGEMM / SpGEAM
                  "matrix_dense_kernels_hip", "matrix_csr_kernels2_hip", "matrix_diagonal_kernels_hip",
                  "stop_residual_norm_kernels_hip", "components_absolute_array_kernels_hip",
                  "components_reduce_array_kernels_hip", "components_precision_conversion_kernels_hip"):
        assert EXPECTED[suite]["known_failures"] == {}, suite
