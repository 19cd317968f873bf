"""GPU parity of the mixed-precision products (csrc/mixed_precision.hip) through the C ABI / the
Python mirror against the oracle (oracle/gko_oracle_mixed.inc, pinned by the live reference built with
GINKGO_MIXED_PRECISION: tests/test_mixed_cpu.py) and against tests/golden/mixed_spmv.npz.
Bar: BIT-EXACT for the real triples (the kernels keep the reference's order: widen, multiply, add in
k order, narrow once).  The complex triples are checked against the live reference in
tests/dropin/mixed_test.cpp (tests/test_dropin_gpu.py::test_mixed_precision_core_flavor)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_mixed_golden as mg  # noqa: E402
from util import random_csr  # noqa: E402

pytestmark = pytest.mark.gpu
DT = mg.DT
TDT = {0: torch.float64, 1: torch.float32}
ALL = [(m, i, o) for m in (0, 1) for i in (0, 1) for o in (0, 1)]


def _bits_equal(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def _csr(g, ex, rp, ci, v, shape):
    return g.Csr.from_arrays(ex, shape, rp, ci, v)


def _ell(g, ex, n_rows, n_cols, k, stride, cols, vals):
    return g.Ell(ex, (n_rows, n_cols), ex.to_device(vals), ex.to_device(cols), k, stride)


@pytest.mark.parametrize("idx", [np.int32, np.int64])
@pytest.mark.parametrize("triple", mg.TRIPLES)
def test_golden_fixture(gexec, triple, idx):
    """the inputs and the live reference's outputs of tests/golden/mixed_spmv.npz, csr and ell"""
    import ginkgo_amd as g
    gold = np.load(os.path.join(ROOT, "tests", "golden", "mixed_spmv.npz"))
    rp, ci, vals, b, c = (gold[k] for k in ("row_ptrs", "col_idxs", "vals", "b", "c"))
    m, i, o = triple
    n_rows, n_cols = len(rp) - 1, b.shape[0]
    v = vals.astype(DT[m])
    for fmt in ("csr", "ell"):
        if fmt == "csr":
            a = _csr(g, gexec, rp.astype(idx), ci.astype(idx), v, (n_rows, n_cols))
        else:
            k, stride, cols, ev = mg.to_ell(rp, ci.astype(idx), v)
            a = _ell(g, gexec, n_rows, n_cols, k, stride, cols, ev)
        key = f"{fmt}_{m}{i}{o}_{'i64' if idx == np.int64 else 'i32'}"
        db = g.Dense.from_numpy(gexec, b.astype(DT[i]))
        dc = g.Dense.from_numpy(gexec, c.astype(DT[o]))
        a.apply(db, dc)
        assert _bits_equal(dc.to_numpy(), gold[key + "_spmv"]), key
        dc = g.Dense.from_numpy(gexec, c.astype(DT[o]))
        a.apply(g.scalar(gexec, float(gold["alpha"]), TDT[m]), db, g.scalar(gexec, float(gold["beta"]), TDT[o]), dc)
        assert _bits_equal(dc.to_numpy(), gold[key + "_adv"]), key + " advanced"


@pytest.mark.parametrize("nrhs", [1, 3])
@pytest.mark.parametrize("triple", ALL)
def test_csr_random_against_the_oracle(gexec, oracle, triple, nrhs):
    """a matrix with empty rows, a row of 3000 entries (several LDS chunks of the plain kernel; the tuned kernels
    keep the reference's order up to GKOC_CSR_LONG_ROW = 4096 entries per row) and a
    last segment of fewer than 64 rows; the uniform triples and (f32, f64, f64) take the tuned
    kernels through the same entry point"""
    import ginkgo_amd as g
    m, i, o = triple
    n_rows, n_cols = 2000 + 37, 6000
    rp, ci, v = random_csr(n_rows, n_cols, 0.004, 11, np.int32, empty_rows=(0, 63, 64, n_rows - 1))
    # one long row
    rng = np.random.default_rng(3)
    long_cols = np.sort(rng.choice(n_cols, 3000, replace=False)).astype(np.int32)
    lens = np.diff(rp)
    r = 1000
    ci = np.concatenate([ci[:rp[r]], long_cols, ci[rp[r + 1]:]])
    v = np.concatenate([v[:rp[r]], rng.uniform(-1, 1, 3000), v[rp[r + 1]:]])
    lens[r] = 3000
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    v = (v * 10.0 ** rng.integers(-2, 3, v.size)).astype(DT[m])
    b = rng.uniform(-1, 1, (n_cols, nrhs)).astype(DT[i])
    c0 = rng.uniform(-1, 1, (n_rows, nrhs)).astype(DT[o])
    db = gexec.to_device(b)
    suf = "gkoc_csr_spmv_mixed_i32"
    from ginkgo_amd._lib import call
    drp, dci, dv = gexec.to_device(rp), gexec.to_device(ci), gexec.to_device(v)
    dc = gexec.to_device(c0.copy())
    call(suf, gexec.stream, C.c_int(m), C.c_int(i), C.c_int(o), n_rows, n_cols, None, drp, dci, dv, db, nrhs, None,
         dc, nrhs, nrhs)
    torch.cuda.synchronize()
    if m == i == o:
        want = oracle.csr_spmv(rp, ci, v, b)
    else:
        want = oracle.csr_spmv_mixed(rp, ci, v, b, DT[o])
    assert _bits_equal(dc.cpu().numpy(), want)
    alpha, beta = np.array([-1.7], DT[m]), np.array([0.3], DT[o])
    dc = gexec.to_device(c0.copy())
    call(suf, gexec.stream, C.c_int(m), C.c_int(i), C.c_int(o), n_rows, n_cols, gexec.to_device(alpha), drp, dci, dv,
         db, nrhs, gexec.to_device(beta), dc, nrhs, nrhs)
    torch.cuda.synchronize()
    if m == i == o:
        want = oracle.csr_spmv(rp, ci, v, b, alpha=alpha[0], beta=beta[0], c=c0)
    else:
        want = oracle.csr_spmv_mixed(rp, ci, v, b, DT[o], alpha=alpha[0], beta=beta[0], c=c0)
    assert _bits_equal(dc.cpu().numpy(), want)


@pytest.mark.parametrize("triple", mg.TRIPLES)
def test_ell_random_against_the_oracle(gexec, oracle, triple):
    m, i, o = triple
    from ginkgo_amd._lib import call
    n_rows, n_cols, nrhs = 3001, 2500, 2
    rp, ci, v = random_csr(n_rows, n_cols, 0.006, 5, np.int64, empty_rows=(7,))
    k, stride, cols, ev = mg.to_ell(rp.astype(np.int64), ci, v.astype(DT[m]))
    rng = np.random.default_rng(9)
    b = rng.uniform(-1, 1, (n_cols, nrhs)).astype(DT[i])
    c0 = rng.uniform(-1, 1, (n_rows, nrhs)).astype(DT[o])
    dcols, dv, db = gexec.to_device(cols), gexec.to_device(ev), gexec.to_device(b)
    dc = gexec.to_device(c0.copy())
    call("gkoc_ell_spmv_mixed_i64", gexec.stream, C.c_int(m), C.c_int(i), C.c_int(o), n_rows, n_cols, k, stride, None,
         dcols, dv, db, nrhs, None, dc, nrhs, nrhs)
    torch.cuda.synchronize()
    assert _bits_equal(dc.cpu().numpy(), oracle.ell_spmv_mixed(n_rows, k, stride, cols, ev, b, DT[o]))
    alpha, beta = np.array([0.9], DT[m]), np.array([-1.1], DT[o])
    dc = gexec.to_device(c0.copy())
    call("gkoc_ell_spmv_mixed_i64", gexec.stream, C.c_int(m), C.c_int(i), C.c_int(o), n_rows, n_cols, k, stride,
         gexec.to_device(alpha), dcols, dv, db, nrhs, gexec.to_device(beta), dc, nrhs, nrhs)
    torch.cuda.synchronize()
    want = oracle.ell_spmv_mixed(n_rows, k, stride, cols, ev, b, DT[o], alpha=alpha[0], beta=beta[0], c=c0)
    assert _bits_equal(dc.cpu().numpy(), want)


def test_row_gather_between_precisions(gexec):
    """dense::row_gather<ValueType, OutputType> (reference/matrix/dense_kernels.cpp:915-925) and the
    advanced form (:931-950: type(alpha * orig) + type(beta) * type(out), type = the wider)"""
    import ginkgo_amd as g
    from ginkgo_amd._lib import call
    rng = np.random.default_rng(1)
    orig = rng.uniform(-1, 1, (500, 7))
    idx = rng.integers(0, 500, 333).astype(np.int32)
    for vt, ot in ((0, 1), (1, 0)):
        src = orig.astype(DT[vt])
        d = g.Dense.from_numpy(gexec, src)
        out = g.Dense.create(gexec, (333, 7), TDT[ot])
        d.row_gather(gexec.to_device(idx), out)
        assert _bits_equal(out.to_numpy(), src[idx].astype(DT[ot]))
        out0 = rng.uniform(-1, 1, (333, 7)).astype(DT[ot])
        dout = gexec.to_device(out0.copy())
        alpha, beta = np.array([1.25], DT[vt]), np.array([-0.3], DT[vt])
        call("gkoc_dense_row_gather_mixed_i32", gexec.stream, C.c_int(vt), C.c_int(ot), 333, 7, gexec.to_device(alpha),
             gexec.to_device(idx), d.values, d.ld, gexec.to_device(beta), dout, 7)
        torch.cuda.synchronize()
        wide = np.float64
        want = ((alpha[0] * src[idx]).astype(wide) + wide(beta[0]) * out0.astype(wide)).astype(DT[ot])
        assert _bits_equal(dout.cpu().numpy(), want)


def test_real_and_complex_in_one_product_is_refused(gexec):
    from ginkgo_amd._lib import NotSupported, call
    z = gexec.zeros((4,), torch.float64)
    zi = gexec.zeros((5,), torch.int32)
    with pytest.raises(NotSupported):
        call("gkoc_csr_spmv_mixed_i32", gexec.stream, C.c_int(0), C.c_int(2), C.c_int(0), 4, 4, None, zi, zi, z, z, 1,
             None, z, 1, 1)
