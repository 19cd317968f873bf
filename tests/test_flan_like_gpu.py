"""configs[4] stand-in.  SuiteSparse Janna/Flan_1565 (3-D mechanical FE model,
3 dof per node, ~73 nnz/row, SPD) is not in the container and there is no
network, so the case is covered by a matrix with the same character: the 27-pt
node graph of a g^3 grid carrying a dense SPD 3x3 block per edge,
A = L27 (x) B  - n = 3 g^3, up to 81 nnz/row, fewer on the boundary, natural 3x3
diagonal blocks.  Checks CSR vs SELL-P vs the oracle and CG + block-Jacobi on
both formats."""
import numpy as np
import pytest
import scipy.sparse as sp

from util import rel_frobenius

pytestmark = pytest.mark.gpu

B3 = np.array([[4.0, 1.0, 0.5], [1.0, 3.0, 0.25], [0.5, 0.25, 2.0]])


def flan_like(oracle, g):
    rp, ci, v = oracle.stencil_csr(3, g)
    n = g ** 3
    l27 = sp.csr_matrix((v, ci, rp), shape=(n, n))
    a = sp.kron(l27, sp.csr_matrix(B3), format="csr")
    a.sort_indices()
    return a


@pytest.mark.parametrize("grid", [6, 14])
def test_flan_like_spmv_formats(gexec, oracle, grid):
    import ginkgo_amd as g
    a = flan_like(oracle, grid)
    n = a.shape[0]
    rp, ci, v = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    lens = np.diff(rp)
    assert lens.max() == 81 and lens.min() == 24          # irregular row lengths
    da = g.Csr.from_scipy(gexec, a)
    x = np.random.default_rng(grid).uniform(-1, 1, n)
    ref = oracle.csr_spmv(rp, ci, v, x)
    y = g.Dense.create(gexec, (n, 1))
    da.apply(g.Dense.from_numpy(gexec, x), y)
    assert np.array_equal(y.to_numpy()[:, 0], ref)
    sl = da.convert_to_sellp()                            # slice_size 64, stride_factor 1
    y2 = g.Dense.create(gexec, (n, 1))
    sl.apply(g.Dense.from_numpy(gexec, x), y2)
    assert np.array_equal(y2.to_numpy()[:, 0], ref)
    # SELL-P pads only to the per-slice maximum: far less than ELL would
    stored = sl.values.numel()
    assert a.nnz <= stored < n * 81


def test_flan_like_cg_block_jacobi(gexec, oracle):
    import ginkgo_amd as g
    grid = 10
    a = flan_like(oracle, grid)
    n = a.shape[0]
    rp, ci, v = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    rhs = np.ones(n)
    xo, iters, _ = oracle.cg_solve(rp, ci, v, rhs, max_iters=2000, reduction=1e-10,
                                   precond="block", max_block_size=3)
    da = g.Csr.from_scipy(gexec, a)
    prec = g.Jacobi.build().with_max_block_size(3).on(gexec).generate(da)
    assert prec.get_num_blocks() == n // 3                # natural 3x3 node blocks
    for op in (da, da.convert_to_sellp()):
        s = (g.Cg.build()
             .with_criteria(g.stop.Iteration.build().with_max_iters(2000),
                            g.stop.ResidualNorm.build().with_reduction_factor(1e-10))
             .with_generated_preconditioner(prec).on(gexec).generate(op))
        x = g.Dense.from_numpy(gexec, np.zeros(n))
        s.apply(g.Dense.from_numpy(gexec, rhs), x)
        assert s.has_converged and abs(s.num_iterations - iters) <= 1
        assert rel_frobenius(x.to_numpy()[:, 0], xo) < 1e-9
        assert np.linalg.norm(rhs - a @ x.to_numpy()[:, 0]) <= 1.01e-10 * np.linalg.norm(rhs)


def test_heavy_tailed_stand_in_formats_and_cg(gexec, oracle):
    """configs[4]'s SECOND stand-in, irregular on purpose (ginkgo_amd/workloads.py irregular_rows): power-law row
    lengths and hub rows - at n = 80 000 one hub has 5 000 entries, beyond GKOC_CSR_LONG_ROW = 4096, where
    the CSR kernel sums the row with the whole wave (csr_spmv_pipe.hpp: partial sums per lane, folded - the
    one place where the row sum is not formed in entry order).  CSR: bit-identical to the oracle on every
    row up to 4096 entries, 1e-15 on the hub rows; SELL-P (lane = row, entry order): bit-identical on ALL rows;
    CG + block-Jacobi on both formats against the oracle's solve.
    Reference: reference/matrix/csr_kernels.cpp:58-94, reference/matrix/sellp_kernels.cpp:27-100."""
    import ginkgo_amd as g
    from ginkgo_amd import workloads as wl
    n = 80000
    rp, ci, v = wl.irregular_rows(n)
    lens = np.diff(rp)
    assert lens.max() > 4096 and np.median(lens) <= 12 and np.percentile(lens, 99.9) < 100     # a heavy tail
    a = sp.csr_matrix((v, ci, rp), shape=(n, n))
    assert abs(a - a.T).max() == 0.0
    da = g.Csr.from_arrays(gexec, (n, n), rp, ci, v)
    x = np.random.default_rng(3).uniform(-1, 1, n)
    ref = oracle.csr_spmv(rp, ci, v, x)
    y = g.Dense.create(gexec, (n, 1))
    da.apply(g.Dense.from_numpy(gexec, x), y)
    got = y.to_numpy()[:, 0]
    short = lens <= 4096
    assert np.array_equal(got[short], ref[short])
    scale = np.abs(a) @ np.abs(x)
    assert np.all(np.abs(got[~short] - ref[~short]) <= 1e-15 * scale[~short] * np.sqrt(lens[~short]))
    # the segments with hub rows are multiplied by 64 workgroups each (csrc/csr_long_rows.hpp): the same bits
    # every time (no floating-point atomics), the tickets are reset for the next product
    first = got.copy()
    for _ in range(3):
        da.apply(g.Dense.from_numpy(gexec, x), y)
        assert np.array_equal(y.to_numpy()[:, 0], first)
    # c = alpha A b + beta c, int64 indices
    c0 = np.random.default_rng(4).uniform(-1, 1, n)
    ref_adv = oracle.csr_spmv(rp, ci, v, x, alpha=-0.75, beta=1.5, c=c0)
    da64 = g.Csr.from_arrays(gexec, (n, n), rp.astype(np.int64), ci.astype(np.int64), v)
    for mat in (da, da64):
        ya = g.Dense.from_numpy(gexec, c0.copy())
        mat.apply(g.scalar(gexec, -0.75), g.Dense.from_numpy(gexec, x), g.scalar(gexec, 1.5), ya)
        ga = ya.to_numpy()[:, 0]
        assert np.array_equal(ga[short], np.ravel(ref_adv)[short])
        assert np.all(np.abs(ga[~short] - np.ravel(ref_adv)[~short]) <=
                      2e-15 * (scale[~short] + np.abs(c0[~short])) * np.sqrt(lens[~short]))
    # the answer does not depend on the switch (GKOC_TUNE_CSR_LONG_ROWS = 0: one wave per long row, as before)
    import ctypes as C
    from ginkgo_amd import _lib
    _lib.call("gkoc_tune_set", C.c_int(12), C.c_int64(0))
    try:
        y0 = g.Dense.create(gexec, (n, 1))
        da.apply(g.Dense.from_numpy(gexec, x), y0)
    finally:
        _lib.call("gkoc_tune_set", C.c_int(12), C.c_int64(1))
    g0 = y0.to_numpy()[:, 0]
    assert np.array_equal(g0[short], got[short])
    assert np.all(np.abs(g0[~short] - got[~short]) <= 2e-15 * scale[~short] * np.sqrt(lens[~short]))
    sl = da.convert_to_sellp()
    y2 = g.Dense.create(gexec, (n, 1))
    sl.apply(g.Dense.from_numpy(gexec, x), y2)
    assert np.array_equal(y2.to_numpy()[:, 0], ref)
    # what the comparison of configs[4] is about: SELL-P pads every 64-row slice to its longest row
    assert sl.values.numel() > 1.3 * a.nnz
    rhs = np.ones(n)
    xo, iters, _ = oracle.cg_solve(rp, ci, v, rhs, max_iters=500, reduction=1e-10, precond="block", max_block_size=4)
    prec = g.Jacobi.build().with_max_block_size(4).on(gexec).generate(da)
    for op in (da, sl):
        s = (g.Cg.build()
             .with_criteria(g.stop.Iteration.build().with_max_iters(500),
                            g.stop.ResidualNorm.build().with_reduction_factor(1e-10))
             .with_generated_preconditioner(prec).on(gexec).generate(op))
        xs = g.Dense.from_numpy(gexec, np.zeros(n))
        s.apply(g.Dense.from_numpy(gexec, rhs), xs)
        assert s.has_converged and abs(s.num_iterations - iters) <= 1, (s.num_iterations, iters)
        assert rel_frobenius(xs.to_numpy()[:, 0], xo) < 1e-9
