"""GPU parity: CSR / ELL / SELL-P SpMV through the C ABI vs the oracle.

Mirrors the reference's cross-executor tests test/matrix/csr_kernels2.cpp:218-470
(532 x 231 random matrix, sorted / unsorted, nrhs 1 and 3, alpha = 2, beta = -1,
strided operands), test/matrix/ell_kernels.cpp, test/matrix/sellp_kernels.cpp and
the known-answer fixtures of reference/test/matrix/csr_kernels.cpp:353-364,
:505-534.  Bar: BIT-EXACT (the kernels keep the reference's summation order).
"""
import numpy as np
import pytest
import torch

from util import random_csr, rel_frobenius

pytestmark = pytest.mark.gpu


def dev_csr(g, ex, rp, ci, v, shape):
    return g.Csr.from_arrays(ex, shape, rp, ci, v)


def test_csr_known_answer(gexec):
    import ginkgo_amd as g
    # reference/test/matrix/csr_kernels.cpp:84-108: [[1,3,2],[0,5,0]]
    a = dev_csr(g, gexec, np.array([0, 3, 4], np.int32),
                np.array([0, 1, 2, 1], np.int32), np.array([1., 3., 2., 5.]), (2, 3))
    x = g.Dense.from_numpy(gexec, np.array([2., 1., 4.]))
    y = g.Dense.create(gexec, (2, 1))
    a.apply(x, y)
    assert y.to_numpy()[:, 0].tolist() == [13.0, 5.0]          # :353-364
    y = g.Dense.from_numpy(gexec, np.array([1., 2.]))
    a.apply(g.scalar(gexec, -1.0), x, g.scalar(gexec, 2.0), y)
    assert y.to_numpy()[:, 0].tolist() == [-11.0, -1.0]        # :505-518
    y = g.Dense.from_numpy(gexec, np.array([np.nan, np.nan]))   # :521-534
    a.apply(g.scalar(gexec, -1.0), x, g.scalar(gexec, 0.0), y)
    assert y.to_numpy()[:, 0].tolist() == [-13.0, -5.0]


@pytest.mark.parametrize("unsorted", [False, True])
@pytest.mark.parametrize("nrhs", [1, 3])
@pytest.mark.parametrize("idx", [np.int32, np.int64])
def test_csr_random_bit_exact(gexec, oracle, unsorted, nrhs, idx):
    import ginkgo_amd as g
    rp, ci, v = random_csr(532, 231, 0.08, 42, idx, unsorted=unsorted,
                           empty_rows=(0, 17, 531))
    rng = np.random.default_rng(7)
    b = rng.uniform(-1, 1, (231, nrhs))
    c0 = rng.uniform(-1, 1, (532, nrhs))
    a = dev_csr(g, gexec, rp, ci, v, (532, 231))
    db = g.Dense.from_numpy(gexec, b)
    dc = g.Dense.create(gexec, (532, nrhs))
    a.apply(db, dc)
    assert np.array_equal(dc.to_numpy(), oracle.csr_spmv(rp, ci, v, b))
    dc = g.Dense.from_numpy(gexec, c0)
    a.apply(g.scalar(gexec, 2.0), db, g.scalar(gexec, -1.0), dc)
    ref = oracle.csr_spmv(rp, ci, v, b, alpha=2.0, beta=-1.0, c=c0)
    assert np.array_equal(dc.to_numpy(), ref)


def test_csr_strided_operands(gexec, oracle):
    import ginkgo_amd as g
    rp, ci, v = random_csr(300, 200, 0.05, 3)
    rng = np.random.default_rng(1)
    b = rng.uniform(-1, 1, (200, 3))
    db = g.Dense.from_numpy(gexec, b, stride=5)
    dc = g.Dense.create(gexec, (300, 3), stride=7)
    dev_csr(g, gexec, rp, ci, v, (300, 200)).apply(db, dc)
    assert np.array_equal(dc.to_numpy(), oracle.csr_spmv(rp, ci, v, b))
    # single strided column view (create_submatrix)
    col = db.create_submatrix((0, 200), (1, 2))
    out = dc.create_submatrix((0, 300), (2, 3))
    dev_csr(g, gexec, rp, ci, v, (300, 200)).apply(col, out)
    assert np.array_equal(out.to_numpy()[:, 0], oracle.csr_spmv(rp, ci, v, b[:, 1].copy()))


def test_csr_f32(gexec, oracle):
    import ginkgo_amd as g
    rp, ci, v = random_csr(400, 400, 0.03, 5, dtype=np.float32)
    b = np.random.default_rng(2).uniform(-1, 1, 400).astype(np.float32)
    dc = g.Dense.create(gexec, (400, 1), torch.float32)
    dev_csr(g, gexec, rp, ci, v, (400, 400)).apply(g.Dense.from_numpy(gexec, b), dc)
    assert np.array_equal(dc.to_numpy()[:, 0], oracle.csr_spmv(rp, ci, v, b))


@pytest.mark.parametrize("idx", [np.int32, np.int64])
def test_mixed_precision_f32_values_f64_vectors(gexec, oracle, idx):
    """csr::spmv / ell::spmv<float, double, double> (arithmetic_type = highest precision): the
    values are stored in float, widened on load, every product and sum is double - the bits of
    the double oracle on the widened values; plain and advanced, several columns, strided
    operands, unsorted columns, rows up to the long-row threshold, empty rows, unaligned views"""
    import ginkgo_amd as g
    rng = np.random.default_rng(17)
    for n_rows, n_cols, dens in ((532, 231, 0.05), (1, 5, 1.0), (300, 300, 0.0), (65, 3000, 0.4), (4100, 70, 0.3)):
        rp, ci, v = random_csr(n_rows, n_cols, dens, n_rows, idx, empty_rows=(0,) if n_rows > 1 else ())
        v32 = v.astype(np.float32)
        wide = v32.astype(np.float64)
        a = dev_csr(g, gexec, rp, ci, v32, (n_rows, n_cols))
        assert a.dtype == torch.float32
        for nrhs in (1, 3):
            b = rng.uniform(-1, 1, (n_cols, nrhs))
            c0 = rng.uniform(-1, 1, (n_rows, nrhs))
            y = g.Dense.from_numpy(gexec, np.full((n_rows, nrhs), np.nan), stride=nrhs + 2)
            a.apply(g.Dense.from_numpy(gexec, b, stride=nrhs + 1), y)
            assert np.array_equal(y.to_numpy(), oracle.csr_spmv(rp, ci, wide, b).reshape(n_rows, nrhs))
            y = g.Dense.from_numpy(gexec, c0)
            a.apply(g.scalar(gexec, 0.7), g.Dense.from_numpy(gexec, b), g.scalar(gexec, -1.3), y)
            want = oracle.csr_spmv(rp, ci, wide, b, alpha=0.7, beta=-1.3, c=c0).reshape(n_rows, nrhs)
            assert np.array_equal(y.to_numpy(), want)
            if n_rows * n_cols == 0 or len(ci) == 0:
                continue
            ell = a.convert_to_ell()
            assert ell.dtype == torch.float32
            y = g.Dense.create(gexec, (n_rows, nrhs))
            ell.apply(g.Dense.from_numpy(gexec, b), y)
            assert np.array_equal(y.to_numpy(), oracle.csr_spmv(rp, ci, wide, b).reshape(n_rows, nrhs))
            y = g.Dense.from_numpy(gexec, c0)
            ell.apply(g.scalar(gexec, 0.7), g.Dense.from_numpy(gexec, b), g.scalar(gexec, -1.3), y)
            assert np.array_equal(y.to_numpy(), want)
    # an unaligned view of the arrays takes the scalar-load instance
    rp, ci, v = random_csr(700, 700, 0.05, 3, idx)
    v32 = v.astype(np.float32)
    vals = gexec.to_device(np.concatenate([[0.0], v32]).astype(np.float32))[1:]
    cols = gexec.to_device(np.concatenate([[0], ci]).astype(idx))[1:]
    a = g.Csr(gexec, (700, 700), vals, cols, gexec.to_device(rp))
    b = rng.uniform(-1, 1, 700)
    y = g.Dense.create(gexec, (700, 1))
    a.apply(g.Dense.from_numpy(gexec, b), y)
    assert np.array_equal(y.to_numpy()[:, 0], oracle.csr_spmv(rp, ci, v32.astype(np.float64), b))
    # the other (matrix, input, output) triples run as well (csrc/mixed_precision.hip,
    # tests/test_mixed_gpu.py) - with the reference's roundings, not a silent conversion of the vectors
    y32 = g.Dense.create(gexec, (700, 1), torch.float32)
    a.apply(g.Dense.from_numpy(gexec, b.astype(np.float32)), y32)
    assert np.array_equal(y32.to_numpy()[:, 0], oracle.csr_spmv(rp, ci, v32, b.astype(np.float32)))
    a64 = dev_csr(g, gexec, rp, ci, v, (700, 700))
    a64.apply(g.Dense.from_numpy(gexec, b.astype(np.float32)), y32)
    assert np.array_equal(y32.to_numpy()[:, 0],
                          oracle.csr_spmv_mixed(rp, ci, v, b.astype(np.float32), np.float32))
    # SELL-P has no mixed instantiations in the reference either (core/matrix/sellp_kernels.hpp)
    with pytest.raises(g.NotSupported):
        a64.convert_to_sellp().apply(g.Dense.from_numpy(gexec, b.astype(np.float32)), y32)


def test_csr_edge_shapes(gexec, oracle):
    import ginkgo_amd as g
    # 0 x 0, all-empty rows, one row, non-multiple-of-64 rows
    a = dev_csr(g, gexec, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0), (0, 0))
    a.apply(g.Dense.create(gexec, (0, 1)), g.Dense.create(gexec, (0, 1)))
    a = dev_csr(g, gexec, np.zeros(6, np.int32), np.zeros(0, np.int32), np.zeros(0), (5, 4))
    y = g.Dense.from_numpy(gexec, np.full(5, np.nan))
    a.apply(g.Dense.from_numpy(gexec, np.ones(4)), y)
    assert y.to_numpy()[:, 0].tolist() == [0.0] * 5
    for rows in (1, 63, 64, 65, 129):
        rp, ci, v = random_csr(rows, 50, 0.3, rows)
        b = np.random.default_rng(rows).uniform(-1, 1, 50)
        y = g.Dense.create(gexec, (rows, 1))
        dev_csr(g, gexec, rp, ci, v, (rows, 50)).apply(g.Dense.from_numpy(gexec, b), y)
        assert np.array_equal(y.to_numpy()[:, 0], oracle.csr_spmv(rp, ci, v, b))


def test_csr_wide_and_long_rows(gexec, oracle):
    """rows whose nnz exceed one LDS tile (1792 products) stay bit-exact (tile
    carry); rows longer than GKOC_CSR_LONG_ROW = 4096 use the cooperative wave
    path and are compared to 1e-14 relative."""
    import ginkgo_amd as g
    rng = np.random.default_rng(11)
    ncols = 9000
    lens = np.array([3000, 10, 0, 2500, 5, 1800] + [40] * 100)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(ncols, l, replace=False)) for l in lens]).astype(np.int32)
    v = rng.uniform(-1, 1, rp[-1])
    b = rng.uniform(-1, 1, ncols)
    y = g.Dense.create(gexec, (len(lens), 1))
    dev_csr(g, gexec, rp, ci, v, (len(lens), ncols)).apply(g.Dense.from_numpy(gexec, b), y)
    assert np.array_equal(y.to_numpy()[:, 0], oracle.csr_spmv(rp, ci, v, b))
    lens = np.array([8000, 3, 5000] + [10] * 70)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(ncols, l, replace=False)) for l in lens]).astype(np.int32)
    v = rng.uniform(-1, 1, rp[-1])
    y = g.Dense.create(gexec, (len(lens), 1))
    dev_csr(g, gexec, rp, ci, v, (len(lens), ncols)).apply(g.Dense.from_numpy(gexec, b), y)
    ref = oracle.csr_spmv(rp, ci, v, b)
    got = y.to_numpy()[:, 0]
    short = lens <= 4096
    assert np.array_equal(got[short], ref[short])
    assert rel_frobenius(got, ref) < 1e-14


@pytest.mark.parametrize("grid", [16, 48])
def test_csr_27pt_stencil(gexec, oracle, grid):
    """the benchmark matrix itself (configs[1] at reduced size): device
    generator index-exact, SpMV bit-exact."""
    import ginkgo_amd as g
    rp, ci, v = oracle.stencil_csr(3, grid)
    a = g.stencil_csr(gexec, 3, grid)
    assert np.array_equal(a.row_ptrs.cpu().numpy(), rp)
    assert np.array_equal(a.col_idxs.cpu().numpy(), ci)
    assert np.array_equal(a.values.cpu().numpy(), v)
    b = np.random.default_rng(42).uniform(-1, 1, grid ** 3)
    y = g.Dense.create(gexec, (grid ** 3, 1))
    a.apply(g.Dense.from_numpy(gexec, b), y)
    assert np.array_equal(y.to_numpy()[:, 0], oracle.csr_spmv(rp, ci, v, b))


def test_stencil_generator_variants(gexec, oracle):
    import ginkgo_amd as g
    for nd, grid, restricted in [(2, 37, True), (2, 20, False), (3, 9, True), (3, 1, False), (2, 1, True)]:
        rp, ci, v = oracle.stencil_csr(nd, grid, restricted)
        a = g.stencil_csr(gexec, nd, grid, restricted)
        assert np.array_equal(a.row_ptrs.cpu().numpy(), rp)
        assert np.array_equal(a.col_idxs.cpu().numpy(), ci)
        assert np.array_equal(a.values.cpu().numpy(), v)
    # z-slab rows == the matching rows of the full matrix
    rp, ci, v = oracle.stencil_csr(3, 10)
    a = g.stencil_csr(gexec, 3, 10, z0=3, nz=4)
    lo, hi = rp[300], rp[700]
    assert np.array_equal(a.row_ptrs.cpu().numpy(), rp[300:701] - lo)
    assert np.array_equal(a.col_idxs.cpu().numpy(), ci[lo:hi])


def test_dimension_mismatch(gexec):
    import ginkgo_amd as g
    a = dev_csr(g, gexec, np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32),
                np.ones(2), (2, 2))
    with pytest.raises(g.DimensionMismatch):
        a.apply(g.Dense.create(gexec, (3, 1)), g.Dense.create(gexec, (2, 1)))
    with pytest.raises(g.DimensionMismatch):
        a.apply(g.Dense.create(gexec, (2, 2)), g.Dense.create(gexec, (2, 1)))


@pytest.mark.parametrize("nrhs", [1, 2, 3, 4, 5, 6, 8, 11])
@pytest.mark.parametrize("idx", [np.int32, np.int64])
def test_ell_bit_exact(gexec, oracle, nrhs, idx):
    import ginkgo_amd as g
    rp, ci, v = random_csr(532, 231, 0.05, 42, idx, empty_rows=(3,))
    a = dev_csr(g, gexec, rp, ci, v, (532, 231))
    rng = np.random.default_rng(3)
    b = rng.uniform(-1, 1, (231, nrhs))
    c0 = rng.uniform(-1, 1, (532, nrhs))
    for stride in (None, 540):
        ell = a.convert_to_ell(stride=stride)
        k, st, ec, ev = oracle.csr_to_ell(rp, ci, v, stride=stride)
        assert (ell.num_stored_per_row, ell.stride) == (k, st)
        assert np.array_equal(ell.col_idxs.cpu().numpy(), ec)
        assert np.array_equal(ell.values.cpu().numpy(), ev)
        y = g.Dense.create(gexec, (532, nrhs))
        ell.apply(g.Dense.from_numpy(gexec, b), y)
        assert np.array_equal(y.to_numpy(), oracle.ell_spmv(532, k, st, ec, ev, b))
        assert np.array_equal(y.to_numpy(), oracle.csr_spmv(rp, ci, v, b))
        y = g.Dense.from_numpy(gexec, c0)
        ell.apply(g.scalar(gexec, 2.0), g.Dense.from_numpy(gexec, b), g.scalar(gexec, -1.0), y)
        assert np.array_equal(y.to_numpy(),
                              oracle.ell_spmv(532, k, st, ec, ev, b, alpha=2.0, beta=-1.0, c=c0))
        # even strides of b and c: three and more columns take the fragment-layout kernel
        # (several lanes per row, csrc/formats.hip), also when the last chunk is partial
        sb, sc = nrhs + nrhs % 2, nrhs + 2 + nrhs % 2
        y = g.Dense.from_numpy(gexec, c0, stride=sc)
        ell.apply(g.Dense.from_numpy(gexec, b, stride=sb), y)
        assert np.array_equal(y.to_numpy(), oracle.csr_spmv(rp, ci, v, b))
        y = g.Dense.from_numpy(gexec, c0, stride=sc)
        ell.apply(g.scalar(gexec, 2.0), g.Dense.from_numpy(gexec, b, stride=sb), g.scalar(gexec, -1.0), y)
        assert np.array_equal(y.to_numpy(),
                              oracle.ell_spmv(532, k, st, ec, ev, b, alpha=2.0, beta=-1.0, c=c0))


@pytest.mark.parametrize("nrhs", [1, 2, 3, 4, 5, 8, 11])
@pytest.mark.parametrize("slice_size,stride_factor", [(64, 1), (32, 2), (2, 2)])
def test_sellp_bit_exact(gexec, oracle, nrhs, slice_size, stride_factor):
    import ginkgo_amd as g
    rp, ci, v = random_csr(532, 231, 0.05, 42, empty_rows=(100,))
    a = dev_csr(g, gexec, rp, ci, v, (532, 231))
    rng = np.random.default_rng(3)
    b = rng.uniform(-1, 1, (231, nrhs))
    c0 = rng.uniform(-1, 1, (532, nrhs))
    sp_ = a.convert_to_sellp(slice_size, stride_factor)
    sets, lens, sc, sv = oracle.csr_to_sellp(rp, ci, v, slice_size, stride_factor)
    assert np.array_equal(sp_.slice_sets.cpu().numpy().view(np.uint64), sets)
    assert np.array_equal(sp_.slice_lengths.cpu().numpy().view(np.uint64), lens)
    assert np.array_equal(sp_.col_idxs.cpu().numpy(), sc)
    assert np.array_equal(sp_.values.cpu().numpy(), sv)
    y = g.Dense.create(gexec, (532, nrhs))
    sp_.apply(g.Dense.from_numpy(gexec, b), y)
    assert np.array_equal(y.to_numpy(), oracle.sellp_spmv(532, slice_size, sets, lens, sc, sv, b))
    assert np.array_equal(y.to_numpy(), oracle.csr_spmv(rp, ci, v, b))
    y = g.Dense.from_numpy(gexec, c0)
    sp_.apply(g.scalar(gexec, 2.0), g.Dense.from_numpy(gexec, b), g.scalar(gexec, -1.0), y)
    assert np.array_equal(
        y.to_numpy(), oracle.sellp_spmv(532, slice_size, sets, lens, sc, sv, b,
                                        alpha=2.0, beta=-1.0, c=c0))
    # even strides: the fragment-layout kernel from three columns on
    ldb, ldc = nrhs + nrhs % 2, nrhs + 2 + nrhs % 2
    y = g.Dense.from_numpy(gexec, c0, stride=ldc)
    sp_.apply(g.scalar(gexec, 2.0), g.Dense.from_numpy(gexec, b, stride=ldb), g.scalar(gexec, -1.0), y)
    assert np.array_equal(
        y.to_numpy(), oracle.sellp_spmv(532, slice_size, sets, lens, sc, sv, b,
                                        alpha=2.0, beta=-1.0, c=c0))


@pytest.mark.parametrize("density,rows", [(0.3, 200), (0.02, 1000)])
def test_setup_kernels_staged_and_fallback(gexec, oracle, density, rows):
    """conversions, sortedness, sort and diagonal extraction go through an LDS stage
    of 2048 elements per 64 rows; density 0.3 x 500 columns (~150 / row) exceeds it
    and takes the direct path, density 0.02 stays inside.  Index-exact / bit-exact."""
    import ginkgo_amd as g
    rp, ci, v = random_csr(rows, 500, density, 17, unsorted=True, empty_rows=(3, 64))
    a = dev_csr(g, gexec, rp, ci.copy(), v.copy(), (rows, 500))
    assert not a.is_sorted_by_column_index()
    a.sort_by_column_index()
    assert a.is_sorted_by_column_index()
    import scipy.sparse as sp
    ref = sp.csr_matrix((v, ci, rp), shape=(rows, 500))
    ref.sort_indices()
    assert np.array_equal(a.col_idxs.cpu().numpy(), ref.indices)
    assert np.array_equal(a.values.cpu().numpy(), ref.data)
    rs, cs, vs = ref.indptr.astype(np.int32), ref.indices.astype(np.int32), ref.data
    d = a.extract_diagonal().cpu().numpy()
    assert np.array_equal(d, oracle.csr_extract_diagonal(rows, 500, rs, cs, vs))
    e = a.convert_to_ell()
    k, stride, ec, ev = oracle.csr_to_ell(rs, cs, vs)
    assert np.array_equal(e.col_idxs.cpu().numpy(), ec) and np.array_equal(e.values.cpu().numpy(), ev)
    sl = a.convert_to_sellp()
    sets, lens, sc, sv = oracle.csr_to_sellp(rs, cs, vs, 64, 1)
    assert np.array_equal(sl.col_idxs.cpu().numpy(), sc) and np.array_equal(sl.values.cpu().numpy(), sv)


@pytest.mark.parametrize("nrhs", [2, 3, 4, 5, 8, 9])
@pytest.mark.parametrize("adv", [False, True])
def test_csr_multi_rhs_single_pass(gexec, oracle, nrhs, adv):
    """several right-hand sides go through the one-pass kernel (csr_spmv_multi.hpp) in
    chunks of 2 or 4 columns: every column bit-identical to the sequential reference
    (and hence to the single-column kernel), for packed and padded strides, aligned and
    odd (scalar b loads), with empty rows and rows spanning several ring passes"""
    import ginkgo_amd as g
    rng = np.random.default_rng(nrhs)
    lens = np.concatenate(([0, 700, 3], rng.integers(0, 40, 700), [0, 0, 1100, 2]))
    rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    n, ncols = len(lens), 1200
    ci = np.concatenate([np.sort(rng.choice(ncols, l, replace=False)) for l in lens]).astype(np.int32)
    v = rng.uniform(-1, 1, len(ci))
    a = dev_csr(g, gexec, rp, ci, v, (n, ncols))
    b = rng.uniform(-1, 1, (ncols, nrhs))
    c0 = rng.uniform(-1, 1, (n, nrhs))
    want = oracle.csr_spmv(rp, ci, v, b, alpha=-0.5, beta=1.5, c=c0) if adv else oracle.csr_spmv(rp, ci, v, b)
    for sb, sc in ((nrhs, nrhs), (nrhs + 1, nrhs + 2), (nrhs + 3, nrhs)):
        db = g.Dense.from_numpy(gexec, b, stride=sb)
        dc = g.Dense.from_numpy(gexec, c0, stride=sc)
        if adv:
            a.apply(g.scalar(gexec, -0.5), db, g.scalar(gexec, 1.5), dc)
        else:
            a.apply(db, dc)
        assert np.array_equal(dc.to_numpy(), want), (sb, sc)


def test_csr_multi_rhs_full_size_columns_match_single(gexec):
    """128^3 27-point stencil, 2..9 right-hand sides: each column of the block product
    equals the single-column product bit for bit; float32 too"""
    import ginkgo_amd as g
    a = g.stencil_csr(gexec, 3, 128)
    n = a.size[0]
    gen = torch.Generator(device="cuda").manual_seed(3)
    for dtype in (torch.float64, torch.float32):
        m = a if dtype == torch.float64 else g.Csr(gexec, a.size, a.values.to(torch.float32), a.col_idxs, a.row_ptrs)
        x = torch.rand((n, 9), generator=gen, device="cuda", dtype=dtype) - 0.5
        single = g.Dense.create(gexec, (n, 9), dtype)
        for j in range(9):
            m.apply(g.Dense(gexec, x[:, j].contiguous().reshape(n, 1)), single.create_submatrix((0, n), (j, j + 1)))
        for k in (2, 3, 4, 8, 9):
            y = g.Dense.create(gexec, (n, k), dtype)
            m.apply(g.Dense(gexec, x[:, :k].contiguous()), y)
            assert torch.equal(y.values[:, :k], single.values[:, :k]), (dtype, k)


def test_arena_places_by_role(gexec):
    """The library's device allocator (csrc/arena.hip) behind gkoc_malloc /
    HipExecutor::raw_alloc: matrix values, index arrays and vectors of >= 1 MiB come from
    regions of different memory classes; contents and results do not depend on it."""
    import ctypes as C
    import ginkgo_amd as g
    from ginkgo_amd import _lib
    a = g.stencil_csr(gexec, 3, 96)                    # values 170 MB, col_idxs 85 MB
    n = a.size[0]
    xh = np.random.default_rng(0).uniform(-1, 1, n)
    x = g.Dense.from_numpy(gexec, xh)
    y = g.Dense.create(gexec, (n, 1))
    a.apply(x, y)
    mode = C.c_int64(0)
    cls = a.memory_classes()
    cy = gexec.memory_class(y.values)
    if cls["values"] >= 0:                              # class regions in use (default)
        assert cls["col_idxs"] >= 0 and cy >= 0
        found = len({cls["values"], cls["col_idxs"], cy})
        assert found >= 2, (cls, cy)                    # at least: output not with the matrix
        assert cy != cls["values"]
    # same numbers as with torch's own allocator
    a2 = g.Csr(gexec, a.size, a.values.clone(), a.col_idxs.clone(), a.row_ptrs.clone())
    y2 = g.Dense(gexec, torch.empty((n, 1), dtype=torch.float64, device=gexec.device))
    a2.apply(g.Dense(gexec, x.values.clone()), y2)
    assert torch.equal(y.values, y2.values)


def test_arena_c_abi(gexec):
    """gkoc_malloc / gkoc_malloc_role / gkoc_free: alignment, no overlap, reuse after
    free, interior pointers refused, gkoc_arena_class_of / gkoc_arena_stats"""
    import ctypes as C
    from ginkgo_amd import _lib
    L = _lib.lib()
    sizes = [100, 4096, 1 << 20, (1 << 20) + 12345, 3 << 20, 64 << 20, 5, 200 << 20]
    roles = [0, 1, 2, 3, 0, 1, 2, 3]
    ptrs = []
    for sz, role in zip(sizes, roles):
        p = C.c_void_p()
        _lib.call("gkoc_malloc_role", C.byref(p), C.c_size_t(sz), C.c_int(role))
        assert p.value and p.value % 256 == 0
        if sz >= (1 << 20):
            assert p.value % (2 << 20) == 0
        ptrs.append((p.value, sz))
    spans = sorted(ptrs)
    for (p0, s0), (p1, _) in zip(spans, spans[1:]):
        assert p0 + s0 <= p1, "allocations overlap"
    # the memory is usable
    t = torch.empty(1)  # noqa: F841 (torch initialised)
    for p, sz in ptrs:
        _lib.call("gkoc_memset", C.c_void_p(p), C.c_int(0x5a), C.c_size_t(sz), gexec.stream)
    gexec.synchronize()
    host = (C.c_uint8 * 100)()
    _lib.call("gkoc_memcpy_d2h", host, C.c_void_p(ptrs[0][0]), C.c_size_t(100), gexec.stream)
    assert bytes(host) == b"\x5a" * 100
    # an interior pointer is not an allocation
    assert L.gkoc_free(C.c_void_p(ptrs[5][0] + 4096)) != 0
    # free and get the same block back
    big = ptrs[5]
    _lib.call("gkoc_free", C.c_void_p(big[0]))
    p = C.c_void_p()
    _lib.call("gkoc_malloc_role", C.byref(p), C.c_size_t(big[1]), C.c_int(1))
    assert p.value == big[0]
    c = C.c_int(-2)
    _lib.call("gkoc_arena_class_of", p, C.byref(c))
    assert -1 <= c.value <= 2
    for q, _ in ptrs:
        _lib.call("gkoc_free", C.c_void_p(q))
    assert L.gkoc_malloc_role(C.byref(p), C.c_size_t(10), C.c_int(7)) != 0      # unknown role


def _tune(key, value):
    import ctypes as C
    from ginkgo_amd import _lib
    _lib.call("gkoc_tune_set", C.c_int(key), C.c_int64(value))


def _short_rows_with_hubs(rng, n, ncols, hubs):
    """a dozen entries per row at most, a few rows of thousands (sorted columns)"""
    lens = rng.integers(0, 14, n)
    for r, l in hubs:
        lens[r] = l
    rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(ncols, l, replace=False)) for l in lens]).astype(np.int32)
    return lens, rp, ci, rng.uniform(-1, 1, rp[-1])


@pytest.mark.parametrize("spw", [0, 1, 2, 4, 8])
def test_csr_short_rows_several_segments_per_wave(gexec, oracle, spw):
    """round 6: matrices with short rows give a wave up to eight 64-row segments (GKOC_TUNE_CSR_SEGS_PER_WAVE,
    0 = the launcher's rule).  Segments with a row beyond GKOC_CSR_LONG_ROW are left to the flagged-segments
    kernel - here they sit at the start, in the middle and at the end of an eight-segment wave, and two of them
    are neighbours - and the ordinary rows next to a hub (170 and 3000 entries among them) keep the reference's
    bits through the products-in-LDS path.  c = A b and c = alpha A b + beta c."""
    import ginkgo_amd as g
    rng = np.random.default_rng(100 + spw)
    n, ncols = 64 * 37 + 11, 20000
    hubs = [(5, 9000), (64 * 3 + 1, 5000), (64 * 3 + 2, 170), (64 * 4 + 9, 4500), (64 * 4 + 10, 3000),
            (64 * 7 + 63, 6000), (64 * 20, 4097), (64 * 20 + 1, 4096), (n - 1, 7000)]
    lens, rp, ci, v = _short_rows_with_hubs(rng, n, ncols, hubs)
    b = rng.uniform(-1, 1, ncols)
    c0 = rng.uniform(-1, 1, n)
    ref = oracle.csr_spmv(rp, ci, v, b)
    ref_adv = np.ravel(oracle.csr_spmv(rp, ci, v, b, alpha=-0.75, beta=1.5, c=c0))
    short = lens <= 4096
    scale = np.abs(sp_csr(rp, ci, v, (n, ncols))) @ np.abs(b)
    _tune(13, spw)
    try:
        a = dev_csr(g, gexec, rp, ci, v, (n, ncols))
        y = g.Dense.create(gexec, (n, 1))
        for _ in range(2):                      # (the second product finds the matrix in the launcher's cache)
            a.apply(g.Dense.from_numpy(gexec, b), y.fill(7.0))
            got = y.to_numpy()[:, 0]
            assert np.array_equal(got[short], ref[short])
            assert np.all(np.abs(got[~short] - ref[~short]) <= 1e-15 * scale[~short] * np.sqrt(lens[~short]))
        ya = g.Dense.from_numpy(gexec, c0.copy())
        a.apply(g.scalar(gexec, -0.75), g.Dense.from_numpy(gexec, b), g.scalar(gexec, 1.5), ya)
        ga = ya.to_numpy()[:, 0]
        assert np.array_equal(ga[short], ref_adv[short])
        assert np.all(np.abs(ga[~short] - ref_adv[~short]) <=
                      2e-15 * (scale[~short] + np.abs(c0[~short])) * np.sqrt(lens[~short]))
    finally:
        _tune(13, 0)


def sp_csr(rp, ci, v, shape):
    import scipy.sparse as sp
    return sp.csr_matrix((v, ci, rp), shape=shape)


def test_csr_short_rows_large_takes_the_rule(gexec, oracle):
    """large enough for the launcher's own rule to hand out several segments per wave (n = 1.5 M rows of 0 .. 13
    entries: 64-row segments of ~400 entries): bit-identical to the oracle"""
    import ginkgo_amd as g
    rng = np.random.default_rng(5)
    n = 1500000
    lens = rng.integers(0, 14, n)
    rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    # banded: columns near the diagonal (cheap to generate, sorted by construction)
    ci = (np.repeat(np.arange(n), lens) + np.concatenate([np.arange(l) for l in lens]) * 3) % n
    order = np.lexsort((ci, np.repeat(np.arange(n), lens)))
    ci = ci[order].astype(np.int32)
    v = rng.uniform(-1, 1, rp[-1])
    b = rng.uniform(-1, 1, n)
    a = dev_csr(g, gexec, rp, ci, v, (n, n))
    y = g.Dense.create(gexec, (n, 1))
    a.apply(g.Dense.from_numpy(gexec, b), y)
    assert np.array_equal(y.to_numpy()[:, 0], oracle.csr_spmv(rp, ci, v, b))


def test_csr_long_rows_two_streams_at_once(gexec, oracle):
    """two products of ONE matrix with hub rows in flight on two streams share nothing that is
    written (the chunk sums are stream-ordered scratch of each launch): both results are right, many times over"""
    import ginkgo_amd as g
    rng = np.random.default_rng(77)
    n, ncols = 64 * 50, 100000
    hubs = [(64 * k + 3, 30000 + 1000 * k) for k in range(0, 50, 5)]
    lens, rp, ci, v = _short_rows_with_hubs(rng, n, ncols, hubs)
    a = dev_csr(g, gexec, rp, ci, v, (n, ncols))
    b1, b2 = rng.uniform(-1, 1, ncols), rng.uniform(-1, 1, ncols)
    d1, d2 = g.Dense.from_numpy(gexec, b1), g.Dense.from_numpy(gexec, b2)
    y1, y2 = g.Dense.create(gexec, (n, 1)), g.Dense.create(gexec, (n, 1))
    a.apply(d1, y1)
    a.apply(d2, y2)
    torch.cuda.synchronize()
    w1, w2 = y1.values.clone(), y2.values.clone()         # one at a time: the answers
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(50):
        y1.fill(0.0)
        y2.fill(0.0)
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            for _ in range(4):
                a.apply(d1, y1)
        with torch.cuda.stream(s2):
            for _ in range(4):
                a.apply(d2, y2)
        torch.cuda.synchronize()
        assert torch.equal(y1.values, w1) and torch.equal(y2.values, w2)
    short = lens <= 4096
    assert np.array_equal(w1.cpu().numpy()[short, 0], oracle.csr_spmv(rp, ci, v, b1)[short])


@pytest.mark.parametrize("variant", [5040, 6040, 7040, 5020])
@pytest.mark.parametrize("nrhs", [3, 4, 8])
def test_csr_multi_rhs_neighbour_reuse_variants(gexec, oracle, variant, nrhs):
    """round 6 (csr_spmv_frag_pipe_kernel NB): a value of b the row below has just gathered is taken from that
    row's lanes - decided on the column indices, so any matrix may come: a banded one (where it applies almost
    everywhere), a random one (almost nowhere), rows of different lengths, empty rows, one past the staging
    capacity; every column bit-identical to the sequential reference"""
    import ginkgo_amd as g
    rng = np.random.default_rng(variant + nrhs)
    n = 16 * 40 + 5
    cases = []
    # banded: row r holds columns r + d for a handful of d (clipped): neighbours share all but one column
    offs = np.array([-40, -39, -38, -1, 0, 1, 38, 39, 40, 200])
    rows, cols = [], []
    for r in range(n):
        c = r + offs
        c = c[(c >= 0) & (c < n)]
        if r % 17 == 3:
            c = c[:0]                              # an empty row
        if r % 11 == 5:
            c = c[::2]                             # a row with holes
        rows.append(np.full(len(c), r))
        cols.append(c)
    lens = np.array([len(c) for c in cols])
    cases.append((lens, np.concatenate(cols)))
    lens2 = np.concatenate(([0, 600, 3], rng.integers(0, 40, n - 5), [0, 2]))
    cases.append((lens2, np.concatenate([np.sort(rng.choice(n, l, replace=False)) for l in lens2])))
    _tune(11, variant)
    try:
        for lens, ci in cases:
            rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
            ci = ci.astype(np.int32)
            v = rng.uniform(-1, 1, len(ci))
            b = rng.uniform(-1, 1, (n, nrhs))
            c0 = rng.uniform(-1, 1, (len(lens), nrhs))
            a = dev_csr(g, gexec, rp, ci, v, (len(lens), n))
            dc = g.Dense.create(gexec, (len(lens), nrhs))
            a.apply(g.Dense.from_numpy(gexec, b), dc)
            assert np.array_equal(dc.to_numpy(), oracle.csr_spmv(rp, ci, v, b))
            dc = g.Dense.from_numpy(gexec, c0)
            a.apply(g.scalar(gexec, -0.5), g.Dense.from_numpy(gexec, b), g.scalar(gexec, 1.5), dc)
            assert np.array_equal(dc.to_numpy(), oracle.csr_spmv(rp, ci, v, b, alpha=-0.5, beta=1.5, c=c0))
    finally:
        _tune(11, 0)
