"""GPU parity of Coo SpMV (apply / apply2, plain and scaled) and the Csr -> Coo / Hybrid
conversions, through the C ABI, against the oracle and
the reference's golden vectors.

Mirrors reference/test/matrix/coo_kernels.cpp (known answers, unsorted input) and
test/matrix/coo_kernels.cpp / hybrid_kernels.cpp (reference vs device on a 532 x 231
random matrix, 1 and 3 right-hand sides, alpha / beta).  Bars: row-sorted input (what
Ginkgo's Coo holds) bit-exact; unsorted input 1e-14 relative (atomics); index arrays of
the conversions bit-exact."""
import os

import numpy as np
import pytest
import torch

from util import random_csr

pytestmark = pytest.mark.gpu

MODES = ("spmv", "advanced_spmv", "spmv2", "advanced_spmv2")


def _coo(g, gexec, n_rows, n_cols, rows, cols, vals):
    return g.Coo(gexec, (n_rows, n_cols), gexec.to_device(vals), gexec.to_device(cols),
                 gexec.to_device(rows))


def _apply(g, gexec, coo, mode, b, alpha, beta, c0):
    x = g.Dense.from_numpy(gexec, c0)
    db = g.Dense.from_numpy(gexec, b)
    al, bt = g.scalar(gexec, alpha, coo.dtype), g.scalar(gexec, beta, coo.dtype)
    if mode == "spmv":
        coo.apply(db, x)
    elif mode == "advanced_spmv":
        coo.apply(al, db, bt, x)
    elif mode == "spmv2":
        coo.apply2(db, x)
    else:
        coo.apply2(al, db, x)
    return x.to_numpy()


def test_known_answers(gexec):
    import ginkgo_amd as g
    rows = np.array([0, 0, 0, 1], np.int32)
    cols = np.array([0, 1, 2, 1], np.int32)
    vals = np.array([1.0, 3.0, 2.0, 5.0])
    x = np.array([[2.0], [1.0], [4.0]])
    for order in ([0, 1, 2, 3], [3, 1, 0, 2]):            # sorted; AppliesToDenseVectorUnsorted
        a = _coo(g, gexec, 2, 3, rows[order], cols[order], vals[order])
        assert np.array_equal(_apply(g, gexec, a, "spmv", x, 1, 0, np.zeros((2, 1)))[:, 0], [13.0, 5.0])
        assert np.array_equal(_apply(g, gexec, a, "advanced_spmv", x, -1.0, 2.0, np.array([[1.0], [2.0]]))[:, 0],
                              [-11.0, -1.0])
        assert np.array_equal(_apply(g, gexec, a, "spmv2", x, 1, 0, np.array([[2.0], [1.0]]))[:, 0], [15.0, 6.0])
        assert np.array_equal(_apply(g, gexec, a, "advanced_spmv2", x, -1.0, 0, np.array([[1.0], [2.0]]))[:, 0],
                              [-12.0, -3.0])


def test_golden_and_unsorted(gexec, oracle):
    import ginkgo_amd as g
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "coo_hybrid.npz"))
    rp, rows, cols, vals, b, c0, perm = (gold[k] for k in ("row_ptrs", "rows", "cols", "vals", "b", "c0", "perm"))
    a = _coo(g, gexec, 532, 231, rows, cols, vals)
    ash = _coo(g, gexec, 532, 231, rows[perm], cols[perm], vals[perm])
    for mode in MODES:
        got = _apply(g, gexec, a, mode, b, 2.0, -1.0, c0)
        assert np.array_equal(got, gold[mode]), mode                       # the reference's own output
        got = _apply(g, gexec, ash, mode, b, 2.0, -1.0, c0)                # unsorted rows: atomics
        want = gold[mode + "_shuffled"]
        assert np.max(np.abs(got - want)) <= 1e-14 * np.max(np.abs(want)), mode
    # Csr -> Coo and Csr -> Hybrid on the device
    csr = g.Csr.from_arrays(gexec, (532, 231), rp, cols, vals)
    coo = csr.convert_to_coo()
    assert np.array_equal(coo.row_idxs.cpu().numpy(), rows)
    for lim in (0, 4, 9, 1000):
        k, st = (int(t) for t in gold[f"hyb{lim}_shape"])
        h = csr.convert_to_hybrid(column_limit=lim)
        assert (h.ell.num_stored_per_row, h.ell.stride) == (k, st)
        for got, name in ((h.ell.col_idxs, "ell_cols"), (h.ell.values, "ell_vals"), (h.coo.row_idxs, "coo_rows"),
                          (h.coo.col_idxs, "coo_cols"), (h.coo.values, "coo_vals")):
            assert np.array_equal(got.cpu().numpy(), gold[f"hyb{lim}_{name}"]), (lim, name)
        y = g.Dense.from_numpy(gexec, np.full((532, 3), np.nan))
        h.apply(g.Dense.from_numpy(gexec, b), y)
        assert np.array_equal(y.to_numpy(), gold[f"hyb{lim}_apply"]), lim
    # the strategy Ginkgo's Hybrid uses unless told otherwise: a quantile of the row lengths
    h = csr.convert_to_hybrid(imbalance_percent=0.8)
    assert h.ell.num_stored_per_row == int(np.sort(np.diff(rp))[int(532 * 0.8)])


@pytest.mark.parametrize("dtype,itype", [(np.float64, np.int32), (np.float64, np.int64), (np.float32, np.int32)])
@pytest.mark.parametrize("nrhs", [1, 3])
def test_equivalence_types_and_shapes(gexec, oracle, dtype, itype, nrhs):
    import ginkgo_amd as g
    rng = np.random.default_rng(nrhs)
    for n_rows, n_cols, dens in ((532, 231, 0.03), (1, 7, 1.0), (300, 300, 0.0), (4100, 50, 0.3)):
        rp, ci, v = random_csr(n_rows, n_cols, dens, seed=n_rows)
        rp, ci, v = rp.astype(itype), ci.astype(itype), v.astype(dtype)
        rows = np.repeat(np.arange(n_rows, dtype=itype), np.diff(rp))
        b = rng.uniform(-1, 1, (n_cols, nrhs)).astype(dtype)
        c0 = rng.uniform(-1, 1, (n_rows, nrhs)).astype(dtype)
        a = _coo(g, gexec, n_rows, n_cols, rows, ci, v)
        for mode in MODES:
            for alpha, beta in ((0.7, -1.3), (2.0, 0.0)):
                want = oracle.coo_apply(mode, n_rows, rows, ci, v, b, alpha, beta, c0)
                got = _apply(g, gexec, a, mode, b, alpha, beta, c0)
                assert np.array_equal(got, want.reshape(got.shape)), (mode, n_rows, alpha, beta)
        # beta = 0 must not read c (NaN-safe, like csr::advanced_spmv)
        got = _apply(g, gexec, a, "advanced_spmv", b, 1.5, 0.0, np.full((n_rows, nrhs), np.nan, dtype=dtype))
        assert np.array_equal(got, oracle.coo_apply("advanced_spmv", n_rows, rows, ci, v, b, 1.5, 0.0,
                                                    np.zeros((n_rows, nrhs), dtype)).reshape(got.shape))
        # conversions: index arrays bit-exact for every limit
        csr = g.Csr.from_arrays(gexec, (n_rows, n_cols), rp, ci, v)
        assert np.array_equal(csr.convert_to_coo().row_idxs.cpu().numpy(), rows)
        for lim in (0, 2, 5):
            lim_eff = min(lim, n_cols)
            ec, ev, crp, cr, cc, cv = oracle.csr_to_hybrid(rp, ci, v, lim_eff, n_rows)
            h = csr.convert_to_hybrid(column_limit=lim)
            assert np.array_equal(h.coo_row_ptrs.cpu().numpy(), crp)
            for got_, want_ in ((h.ell.col_idxs, ec), (h.ell.values, ev), (h.coo.row_idxs, cr),
                                (h.coo.col_idxs, cc), (h.coo.values, cv)):
                assert np.array_equal(got_.cpu().numpy(), want_), (n_rows, lim)


@pytest.mark.parametrize("dtype,itype", [(np.float64, np.int32), (np.float64, np.int64), (np.float32, np.int32)])
def test_one_pass_operations_that_read_c(gexec, oracle, dtype, itype):
    """From 16 entries per row on, the operations that read c also run in one pass and keep a
    copy of c for the atomic fallback: sorted input bit-exact for every (alpha, beta), unsorted
    input (the kernel has overwritten c by the time it knows) restored and redone with atomics,
    empty leading / trailing rows, and both toggles of the switch give the same bits"""
    import ginkgo_amd as g
    import ctypes as C
    from ginkgo_amd import _lib
    rng = np.random.default_rng(11)
    for n_rows, n_cols, dens in ((700, 90, 0.4), (9000, 64, 0.5), (65, 2000, 0.05)):
        rp, ci, v = random_csr(n_rows, n_cols, dens, seed=n_rows)
        keep = np.ones(len(ci), bool)
        for r in list(range(0, 70)) + list(range(n_rows - 3, n_rows)):
            if n_rows > 100:
                keep[rp[r]:rp[r + 1]] = False            # 70 empty rows in front, 3 at the end
        rows = np.repeat(np.arange(n_rows, dtype=itype), np.diff(rp))[keep]
        ci, v = ci[keep].astype(itype), v[keep].astype(dtype)
        assert len(ci) >= 16 * n_rows
        b = rng.uniform(-1, 1, (n_cols, 1)).astype(dtype)
        c0 = rng.uniform(-1, 1, (n_rows, 1)).astype(dtype)
        a = _coo(g, gexec, n_rows, n_cols, rows, ci, v)
        perm = rng.permutation(len(ci))
        ash = _coo(g, gexec, n_rows, n_cols, rows[perm], ci[perm], v[perm])
        for mode in MODES:
            for alpha, beta in ((0.7, -1.3), (2.0, 0.0), (1.0, 1.0)):
                want = oracle.coo_apply(mode, n_rows, rows, ci, v, b, alpha, beta, c0)
                got = _apply(g, gexec, a, mode, b, alpha, beta, c0)
                assert np.array_equal(got, want.reshape(got.shape)), (mode, n_rows, alpha, beta)
                _lib.call("gkoc_tune_set", C.c_int(4), C.c_int64(0))
                try:
                    two_pass = _apply(g, gexec, a, mode, b, alpha, beta, c0)
                finally:
                    _lib.call("gkoc_tune_set", C.c_int(4), C.c_int64(1))
                assert np.array_equal(got, two_pass), (mode, n_rows, alpha, beta)
                got = _apply(g, gexec, ash, mode, b, alpha, beta, c0)
                tol = (1e-13 if dtype == np.float64 else 1e-5) * np.max(np.abs(want))
                assert np.max(np.abs(got - want.reshape(got.shape))) <= tol, (mode, n_rows, alpha, beta)
        got = _apply(g, gexec, a, "advanced_spmv", b, 1.5, 0.0, np.full((n_rows, 1), np.nan, dtype=dtype))
        assert not np.isnan(got).any()


def test_27pt_full_formats_agree(gexec, oracle):
    """Coo and Hybrid of the 27-pt Laplacian give the bits of the Csr product"""
    import ginkgo_amd as g
    grid = 48
    a = g.stencil_csr(gexec, 3, grid)
    n = grid ** 3
    x = g.Dense.from_numpy(gexec, np.random.default_rng(4).uniform(-1, 1, n))
    y0, y1, y2 = (g.Dense.create(gexec, (n, 1)) for _ in range(3))
    a.apply(x, y0)
    a.convert_to_coo().apply(x, y1)
    a.convert_to_hybrid(column_limit=18).apply(x, y2)
    assert np.array_equal(y0.to_numpy(), y1.to_numpy())
    assert np.array_equal(y0.to_numpy(), y2.to_numpy())


@pytest.mark.parametrize("dtype,itype", [(np.float64, np.int32), (np.float64, np.int64), (np.float32, np.int32)])
def test_transpose(gexec, oracle, dtype, itype):
    """csr::transpose: the reference's result is the stable sort of the entries by column;
    index arrays and values bit-exact, incl. unsorted / duplicate columns inside a row,
    empty rows and columns, rectangular shapes"""
    import ginkgo_amd as g
    for n_rows, n_cols, dens, seed in ((532, 231, 0.03, 1), (3, 4000, 0.2, 2), (4000, 3, 0.4, 3),
                                       (50, 50, 0.0, 4), (1000, 1000, 0.01, 5)):
        rp, ci, v = random_csr(n_rows, n_cols, dens, seed=seed)
        rng = np.random.default_rng(seed)
        ci = ci.copy()
        for r in range(0, n_rows, 3):                 # unsorted and duplicated columns
            seg = slice(rp[r], rp[r + 1])
            if rp[r + 1] - rp[r] > 1:
                ci[seg] = rng.permutation(ci[seg])
                ci[rp[r]] = ci[rp[r] + 1]
        rp, ci, v = rp.astype(itype), ci.astype(itype), v.astype(dtype)
        a = g.Csr.from_arrays(gexec, (n_rows, n_cols), rp, ci, v)
        t = a.transpose()
        trp, tc, tv = oracle.csr_transpose(n_rows, n_cols, rp, ci, v)
        assert t.size == (n_cols, n_rows)
        assert np.array_equal(t.row_ptrs.cpu().numpy(), trp)
        assert np.array_equal(t.col_idxs.cpu().numpy(), tc)
        assert np.array_equal(t.values.cpu().numpy(), tv)
    # 27-pt Laplacian is symmetric: A^T x and A x agree bit for bit (same entries per row,
    # same order since columns are sorted)
    a = g.stencil_csr(gexec, 3, 40)
    x = g.Dense.from_numpy(gexec, np.random.default_rng(0).uniform(-1, 1, 40 ** 3))
    y0, y1 = g.Dense.create(gexec, (40 ** 3, 1)), g.Dense.create(gexec, (40 ** 3, 1))
    a.apply(x, y0)
    a.transpose().apply(x, y1)
    assert np.array_equal(y0.to_numpy(), y1.to_numpy())


@pytest.mark.parametrize("itype", [np.int32, np.int64])
def test_row_pointer_pass_with_gaps(gexec, oracle, itype):
    """pass 1 (row_idxs -> row pointers in one kernel): runs of empty rows of every length around
    the lane-fills-it (8) and wave-fills-it thresholds, in front of the first and behind the last
    stored row, several long gaps inside one lane's four entries, a single stored entry"""
    import ginkgo_amd as g
    rng = np.random.default_rng(5)
    cases = []
    # (n_rows, stored rows)
    cases.append((5000, np.array([4999])))                       # one entry, 4999 empty rows in front
    cases.append((5000, np.array([0])))                          # ... behind
    cases.append((9000, np.array([700, 701, 1500, 1509, 1510, 1519, 8000])))
    gaps = np.array([1, 2, 7, 8, 9, 10, 63, 64, 65, 66, 127, 128, 129, 300, 5, 1000, 3, 2000])
    cases.append((int(gaps.sum()) + 500, np.cumsum(gaps)))
    dense_rows = np.sort(rng.choice(20000, 6000, replace=False))
    cases.append((20011, dense_rows))
    for n_rows, stored in cases:
        n_cols = 37
        counts = rng.integers(1, 4, len(stored))
        rows = np.repeat(stored, counts).astype(itype)
        cols = rng.integers(0, n_cols, len(rows)).astype(itype)
        vals = rng.uniform(-1, 1, len(rows))
        b = rng.uniform(-1, 1, (n_cols, 1))
        c0 = rng.uniform(-1, 1, (n_rows, 1))
        a = _coo(g, gexec, n_rows, n_cols, rows, cols, vals)
        for mode in MODES:
            want = oracle.coo_apply(mode, n_rows, rows, cols, vals, b, 0.7, -1.3, c0)
            got = _apply(g, gexec, a, mode, b, 0.7, -1.3, c0)
            assert np.array_equal(got, want.reshape(got.shape)), (mode, n_rows)
