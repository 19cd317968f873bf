"""The oracle's restatement of the mixed-precision products (oracle/gko_oracle_mixed.inc: csr / ell
spmv + advanced_spmv for the six non-uniform (matrix, input, output) triples of float32 / float64)
against (a) the known answers of the reference's own tests, (b) tests/golden/mixed_spmv.npz, written
by the live reference built with GINKGO_MIXED_PRECISION (tests/golden/make_mixed_golden.py), and
(c) that live reference itself on fresh inputs when oracle/_ref/mixed is present."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from oracle import gko_oracle as o  # noqa: E402
import make_mixed_golden as mg  # noqa: E402

DT = mg.DT


@pytest.mark.parametrize("triple", mg.TRIPLES)
def test_known_answers_of_the_reference_tests(triple):
    """reference/test/matrix/csr_kernels.cpp:367-411 (MixedAppliesToDenseVector1-3: y = (13, 5)),
    :537-589 (MixedAppliesLinearCombinationToDenseVector1-3: y = (-11, -1)) and the same vectors on
    the Ell fixture of reference/test/matrix/ell_kernels.cpp:90-135, 262-310 - for EVERY triple, not
    only the three the typed tests instantiate"""
    m, i, ot = triple
    rp = np.array([0, 3, 4], np.int32)
    ci = np.array([0, 1, 2, 1], np.int32)
    v = np.array([1, 3, 2, 5], DT[m])
    x = np.array([2, 1, 4], DT[i])
    assert o.csr_spmv_mixed(rp, ci, v, x, DT[ot]).tolist() == [13.0, 5.0]
    y = o.csr_spmv_mixed(rp, ci, v, x, DT[ot], alpha=-1.0, beta=2.0, c=np.array([1, 2], DT[ot]))
    assert y.tolist() == [-11.0, -1.0] and y.dtype == DT[ot]
    k, stride, cols, ev = mg.to_ell(rp.astype(np.int64), ci, v)
    assert o.ell_spmv_mixed(2, k, stride, cols, ev, x, DT[ot]).tolist() == [13.0, 5.0]
    y = o.ell_spmv_mixed(2, k, stride, cols, ev, x, DT[ot], alpha=-1.0, beta=2.0, c=np.array([1, 2], DT[ot]))
    assert y.tolist() == [-11.0, -1.0]


@pytest.mark.parametrize("idx64", [False, True])
@pytest.mark.parametrize("fmt", [0, 1])
def test_oracle_reproduces_the_golden_fixture(fmt, idx64):
    g = np.load(os.path.join(ROOT, "tests", "golden", "mixed_spmv.npz"))
    rp, ci, vals, b, c = (g[k] for k in ("row_ptrs", "col_idxs", "vals", "b", "c"))
    for t in mg.TRIPLES:
        key = f"{('csr', 'ell')[fmt]}_{t[0]}{t[1]}{t[2]}_{'i64' if idx64 else 'i32'}"
        got = mg.oracle(fmt, t, idx64, rp, ci, vals, b, c)
        want = g[key + "_spmv"]
        assert got.dtype == want.dtype and np.array_equal(got.view(np.uint8), want.view(np.uint8)), key
        got = mg.oracle(fmt, t, idx64, rp, ci, vals, b, c, float(g["alpha"]), float(g["beta"]))
        want = g[key + "_adv"]
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), key
    # the roundings are really exercised: the float and the double result of one product differ
    a = g["csr_011_i32_spmv"].astype(np.float64)
    bb = g["csr_010_i32_spmv"]
    assert not np.array_equal(a, bb)


def test_oracle_against_the_live_mixed_precision_reference():
    if not os.path.exists(mg.SHIM):
        pytest.skip("oracle/_ref/mixed not built (python oracle/build_ref_mixed.py)")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_mixed_golden.py"), "--check"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "oracle == live reference" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
