"""Block-Jacobi(8) apply with several right-hand sides on the f64 matrix cores
(jacobi_apply_mfma_kernel; GKOC_TUNE_JACOBI_MFMA: 0 never, 1 from two columns, 2 = default from
nine, 3 from four): against the oracle's apply_block (reference/preconditioner/jacobi_kernels.cpp:
419-531) with the tolerance of fused multiply-adds (1e-14 relative; measured 5e-16) - everything
else in the Jacobi path, including two to eight columns (jacobi_apply_fixed_multi_kernel, round 3),
is compared bit for bit."""
import ctypes as C

import numpy as np
import pytest

from test_krylov_gpu import _block_matrix

pytestmark = pytest.mark.gpu

GKOC_TUNE_JACOBI_MFMA = 3


@pytest.fixture
def mfma_on():
    from ginkgo_amd import _lib
    _lib.call("gkoc_tune_set", C.c_int(GKOC_TUNE_JACOBI_MFMA), C.c_int64(1))
    yield
    _lib.call("gkoc_tune_set", C.c_int(GKOC_TUNE_JACOBI_MFMA), C.c_int64(2))


@pytest.mark.parametrize("nrhs", [2, 5, 16, 19, 33])
def test_mfma_apply_matches_the_oracle(gexec, oracle, mfma_on, nrhs):
    import ginkgo_amd as g
    rng = np.random.default_rng(nrhs)
    # block sizes 1..8, a number of blocks that leaves the last storage group partly filled
    sizes = rng.integers(1, 9, 203)
    a = _block_matrix(8, sizes)
    n = a.shape[0]
    rp, ci, v = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    da = g.Csr.from_arrays(gexec, (n, n), rp, ci, v)
    jac = g.Jacobi.build().with_max_block_size(8).on(gexec).generate(da)
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, 8)
    scheme = oracle.jacobi_storage_scheme(8)
    blocks = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
    assert np.array_equal(jac.blocks.cpu().numpy(), blocks)          # generate is untouched
    b = rng.uniform(-1, 1, (n, nrhs))
    x0 = rng.uniform(-1, 1, (n, nrhs))
    want = oracle.jacobi_apply(nb, scheme, ptrs, blocks, b)
    scale = np.max(np.abs(want))
    # simple_apply, padded strides
    x = g.Dense.from_numpy(gexec, np.full((n, nrhs), np.nan), nrhs + 3)
    jac.apply(g.Dense.from_numpy(gexec, b, nrhs + 1), x)
    got = x.to_numpy()
    assert np.max(np.abs(got - want)) <= 1e-14 * scale
    assert not np.array_equal(got, want) or nrhs < 3      # FMA: some bits differ (sanity: the path ran)
    # advanced apply: x = alpha M b + beta x, and beta = 0 must not read x
    x = g.Dense.from_numpy(gexec, x0)
    jac.apply(g.scalar(gexec, 2.0), g.Dense.from_numpy(gexec, b), g.scalar(gexec, -1.0), x)
    want2 = oracle.jacobi_apply(nb, scheme, ptrs, blocks, b, 2.0, -1.0, x0)
    assert np.max(np.abs(x.to_numpy() - want2)) <= 1e-14 * np.max(np.abs(want2))
    x = g.Dense.from_numpy(gexec, np.full((n, nrhs), np.nan))
    jac.apply(g.scalar(gexec, 0.5), g.Dense.from_numpy(gexec, b), g.scalar(gexec, 0.0), x)
    assert np.max(np.abs(x.to_numpy() - 0.5 * want)) <= 1e-14 * scale


def test_single_column_and_other_layouts_keep_the_exact_kernels(gexec, oracle, mfma_on):
    """one right-hand side and block sizes other than 8 never take the matrix-core path"""
    import ginkgo_amd as g
    rng = np.random.default_rng(3)
    for max_bs, nrhs in ((8, 1), (4, 3), (16, 3)):
        sizes = rng.integers(1, min(max_bs, 8) + 1, 50)
        a = _block_matrix(max_bs, sizes)
        n = a.shape[0]
        rp, ci, v = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
        jac = g.Jacobi.build().with_max_block_size(max_bs).on(gexec).generate(
            g.Csr.from_arrays(gexec, (n, n), rp, ci, v))
        nb, ptrs = oracle.jacobi_find_blocks(rp, ci, max_bs)
        scheme = oracle.jacobi_storage_scheme(max_bs)
        blocks = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
        b = rng.uniform(-1, 1, (n, nrhs))
        x = g.Dense.create(gexec, (n, nrhs))
        jac.apply(g.Dense.from_numpy(gexec, b), x)
        assert np.array_equal(x.to_numpy(), oracle.jacobi_apply(nb, scheme, ptrs, blocks, b))


def test_default_is_exact_up_to_eight_columns_and_matrix_cores_from_nine(gexec, oracle):
    import ginkgo_amd as g
    rng = np.random.default_rng(11)
    a = _block_matrix(8, rng.integers(1, 9, 90))
    n = a.shape[0]
    rp, ci, v = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    jac = g.Jacobi.build().with_max_block_size(8).on(gexec).generate(g.Csr.from_arrays(gexec, (n, n), rp, ci, v))
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, 8)
    scheme = oracle.jacobi_storage_scheme(8)
    blocks = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
    for nrhs in (1, 2, 3, 4, 7, 8, 9, 12):
        b = rng.uniform(-1, 1, (n, nrhs))
        x = g.Dense.create(gexec, (n, nrhs))
        jac.apply(g.Dense.from_numpy(gexec, b), x)
        want = oracle.jacobi_apply(nb, scheme, ptrs, blocks, b)
        if nrhs <= 8:
            assert np.array_equal(x.to_numpy(), want), nrhs
        else:
            assert np.max(np.abs(x.to_numpy() - want)) <= 1e-14 * np.max(np.abs(want)), nrhs


@pytest.mark.parametrize("max_bs", [2, 8, 16])
@pytest.mark.parametrize("nrhs", [2, 3, 4, 5, 8])
def test_multi_column_kernel_is_bit_identical(gexec, oracle, max_bs, nrhs):
    """two to eight columns in the fast-path layout: the blocks stay in registers, the lanes load
    their own rows of b (pairs when the strides allow) - simple and advanced apply, packed, even and
    odd strides, beta = 0 over NaNs, a last storage group that is partly filled"""
    import ginkgo_amd as g
    rng = np.random.default_rng(100 * max_bs + nrhs)
    sizes = rng.integers(1, min(max_bs, 8) + 1, 131)
    a = _block_matrix(max_bs, sizes)
    n = a.shape[0]
    rp, ci, v = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    jac = g.Jacobi.build().with_max_block_size(max_bs).on(gexec).generate(
        g.Csr.from_arrays(gexec, (n, n), rp, ci, v))
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, max_bs)
    scheme = oracle.jacobi_storage_scheme(max_bs)
    blocks = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
    b = rng.uniform(-1, 1, (n, nrhs))
    x0 = rng.uniform(-1, 1, (n, nrhs))
    want = oracle.jacobi_apply(nb, scheme, ptrs, blocks, b)
    want2 = oracle.jacobi_apply(nb, scheme, ptrs, blocks, b, 2.0, -1.0, x0)
    for sb, sx in ((nrhs, nrhs), (nrhs + nrhs % 2, nrhs + 2 + nrhs % 2), (nrhs + 1, nrhs + 3)):
        x = g.Dense.from_numpy(gexec, np.full((n, nrhs), np.nan), sx)
        jac.apply(g.Dense.from_numpy(gexec, b, sb), x)
        assert np.array_equal(x.to_numpy(), want), (sb, sx)
        x = g.Dense.from_numpy(gexec, x0, sx)
        jac.apply(g.scalar(gexec, 2.0), g.Dense.from_numpy(gexec, b, sb), g.scalar(gexec, -1.0), x)
        assert np.array_equal(x.to_numpy(), want2), (sb, sx)
        x = g.Dense.from_numpy(gexec, np.full((n, nrhs), np.nan), sx)
        jac.apply(g.scalar(gexec, 0.5), g.Dense.from_numpy(gexec, b, sb), g.scalar(gexec, 0.0), x)
        assert np.array_equal(x.to_numpy(), oracle.jacobi_apply(nb, scheme, ptrs, blocks, b, 0.5, 0.0, x0)), (sb, sx)
