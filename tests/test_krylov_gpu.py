"""GPU parity: Dense BLAS-1, fused CG steps, stopping kernels, (block-)Jacobi
and the end-to-end CG solve, all through the C ABI, vs the oracle.

Mirrors test/matrix/dense_kernels.cpp, test/solver/cg_kernels.cpp:106-191
(597 x 43 with strides 45/46, a stopped column, zero prev_rho / beta),
test/stop/residual_norm_kernels.cpp, test/preconditioner/jacobi_kernels.cpp and
reference/test/solver/cg_kernels.cpp:215-226, :407-424 (known answers).
Bars: element-wise kernels, block pointers, inverse blocks, Jacobi apply:
bit-exact.  Reductions (dot / norm2): |err| <= 1e-13 relative (different
summation tree).  Full solves: relative error <= 1e-9, iteration count +-1.
"""
import numpy as np
import pytest
import torch

from util import random_csr, rel_frobenius

pytestmark = pytest.mark.gpu

RED_TOL = 1e-13


# ------------------------------------------------------------------- dense
@pytest.mark.parametrize("rows,cols,stride", [(597, 1, 1), (597, 43, 45), (10000, 3, 3), (1, 7, 9), (4097, 1, 1)])
def test_dense_elementwise_bit_exact(gexec, oracle, rows, cols, stride):
    import ginkgo_amd as g
    rng = np.random.default_rng(rows + cols)
    x = rng.uniform(-1, 1, (rows, cols))
    y = rng.uniform(-1, 1, (rows, cols))
    for alpha in (np.array([[0.7]]), np.array([[0.0]]), rng.uniform(-1, 1, (1, cols))):
        da = g.Dense.from_numpy(gexec, alpha)
        a1 = alpha.reshape(-1)
        dx = g.Dense.from_numpy(gexec, x, stride)
        assert np.array_equal(dx.scale(da).to_numpy(), oracle.dense_scale(a1, x))
        if np.all(a1 != 0):
            dx = g.Dense.from_numpy(gexec, x, stride)
            assert np.array_equal(dx.inv_scale(da).to_numpy(), oracle.dense_scale(a1, x, inverse=True))
        dy = g.Dense.from_numpy(gexec, y, stride + 1)
        dx = g.Dense.from_numpy(gexec, x, stride)
        assert np.array_equal(dy.add_scaled(da, dx).to_numpy(), oracle.dense_add_scaled(a1, x, y))
        dy = g.Dense.from_numpy(gexec, y, stride + 1)
        assert np.array_equal(dy.sub_scaled(da, dx).to_numpy(), oracle.dense_add_scaled(a1, x, y, subtract=True))
    d = g.Dense.create(gexec, (rows, cols), stride=stride).fill(3.25)
    assert np.all(d.to_numpy() == 3.25)
    assert np.array_equal(g.Dense.create(gexec, (rows, cols)).copy_from(g.Dense.from_numpy(gexec, x, stride)).to_numpy(), x)


def test_dense_unaligned_views(gexec, oracle):
    """sub-views are only 8-byte aligned: the 16-byte vector path must not be
    taken (GMRES Krylov columns, create_submatrix)."""
    import ginkgo_amd as g
    rng = np.random.default_rng(0)
    big = rng.uniform(-1, 1, (2001, 1))
    d = g.Dense.from_numpy(gexec, big)
    v = d.create_submatrix((1, 2000), (0, 1))       # offset by one double
    w = g.Dense.from_numpy(gexec, big).create_submatrix((1, 2000), (0, 1))
    a = g.scalar(gexec, 0.3)
    assert np.array_equal(v.add_scaled(a, w).to_numpy(), oracle.dense_add_scaled([0.3], big[1:2000], big[1:2000]))
    res = g.Dense.create(gexec, (1, 1))
    w.compute_dot(w, res)
    ref = oracle.dense_dot(big[1:2000], big[1:2000])
    assert abs(res.to_numpy()[0, 0] - ref[0]) <= RED_TOL * abs(ref[0])


@pytest.mark.parametrize("rows,cols,stride", [(597, 1, 1), (597, 43, 46), (100003, 1, 1), (2 ** 20, 1, 1), (50000, 5, 8), (3, 2000, 2000), (0, 3, 3)])
def test_dense_reductions(gexec, oracle, rows, cols, stride):
    import ginkgo_amd as g
    rng = np.random.default_rng(rows * 7 + cols)
    x = rng.uniform(-1, 1, (rows, cols))
    y = rng.uniform(-1, 1, (rows, cols))
    dx, dy = g.Dense.from_numpy(gexec, x, stride), g.Dense.from_numpy(gexec, y, stride)
    res = g.Dense.create(gexec, (1, cols))
    scale = np.sum(np.abs(x * y), axis=0) + 1e-300
    got = dx.compute_dot(dy, res).to_numpy()[0]
    assert np.all(np.abs(got - oracle.dense_dot(x, y)) <= RED_TOL * scale)
    got2 = dx.compute_dot(dy, res).to_numpy()[0]
    assert np.array_equal(got, got2), "reduction must be deterministic"
    got = dx.compute_norm2(res).to_numpy()[0]
    ref = oracle.dense_norm2(x)
    assert np.all(np.abs(got - ref) <= RED_TOL * (ref + 1e-300))
    got = dx.compute_squared_norm2(res).to_numpy()[0]
    ref = oracle.dense_norm2(x, squared=True)
    assert np.all(np.abs(got - ref) <= RED_TOL * (ref + 1e-300))


def test_dense_f32(gexec, oracle):
    import ginkgo_amd as g
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, (70001, 1)).astype(np.float32)
    y = rng.uniform(-1, 1, (70001, 1)).astype(np.float32)
    dx, dy = g.Dense.from_numpy(gexec, x), g.Dense.from_numpy(gexec, y)
    a = g.Dense.from_numpy(gexec, np.array([[0.5]], np.float32))
    assert np.array_equal(dy.add_scaled(a, dx).to_numpy(), oracle.dense_add_scaled(np.float32([0.5]), x, y))
    res = g.Dense.create(gexec, (1, 1), torch.float32)
    ref = np.dot(x[:, 0].astype(np.float64), x[:, 0].astype(np.float64))
    assert abs(dx.compute_dot(dx, res).to_numpy()[0, 0] - ref) <= 1e-5 * ref


def test_row_gather(gexec):
    import ginkgo_amd as g
    rng = np.random.default_rng(4)
    x = rng.uniform(-1, 1, (500, 3))
    for dt in (np.int32, np.int64):
        idx = rng.integers(0, 500, 77).astype(dt)
        out = g.Dense.create(gexec, (77, 3), stride=4)
        g.Dense.from_numpy(gexec, x, 5).row_gather(gexec.to_device(idx), out)
        assert np.array_equal(out.to_numpy(), x[idx])


# ---------------------------------------------------------------- CG steps
def _cg_data(rows, cols, seed):
    rng = np.random.default_rng(seed)
    m = lambda: rng.uniform(-1, 1, (rows, cols))
    s = lambda: rng.uniform(0.1, 1, cols)
    return rng, m, s


@pytest.mark.parametrize("rows,cols,stride", [(597, 43, 45), (100000, 1, 1), (1023, 1, 1)])
def test_cg_steps_bit_exact(gexec, oracle, rows, cols, stride):
    import ginkgo_amd as g
    from ginkgo_amd._lib import call
    rng, m, s = _cg_data(rows, cols, 31)
    b, p, z, x, r, q = m(), m(), m(), m(), m(), m()
    rho, prev_rho, beta = s(), s(), s()
    stop = np.zeros(cols, np.uint8)
    if cols > 2:
        stop[1] = 0x81          # one stopped (converged) column
        prev_rho[2] = 0.0       # zero prev_rho: p = z
        beta[0] = 0.0           # zero beta: no update
    ex = gexec
    D = lambda a, st=stride: g.Dense.from_numpy(ex, a, st)
    dv = lambda a: ex.to_device(a)
    # initialize
    db, dr, dz, dp, dq = D(b), D(m()), D(m()), D(m(), stride + 1), D(m())
    d_prev, d_rho, d_stop = dv(np.full(cols, np.nan)), dv(np.full(cols, np.nan)), dv(np.full(cols, 0xFF, np.uint8))
    call("gkoc_cg_initialize_f64", ex.stream, rows, cols, db.values, db.ld, dr.values, dr.ld,
         dz.values, dz.ld, dp.values, dp.ld, dq.values, dq.ld, d_prev, d_rho, d_stop)
    o = oracle.cg_initialize(b)
    for got, ref in zip((dr, dz, dp, dq), o[:4]):
        assert np.array_equal(got.to_numpy(), ref)
    assert np.array_equal(d_prev.cpu().numpy(), o[4]) and np.array_equal(d_rho.cpu().numpy(), o[5])
    assert np.array_equal(d_stop.cpu().numpy(), o[6])
    # step_1
    dp, dz = D(p, stride + 1), D(z)
    call("gkoc_cg_step_1_f64", ex.stream, rows, cols, dp.values, dp.ld, dz.values, dz.ld,
         dv(rho), dv(prev_rho), dv(stop))
    assert np.array_equal(dp.to_numpy(), oracle.cg_step_1(p, z, rho, prev_rho, stop))
    # step_2
    dx, dr, dp, dq = D(x), D(r, stride + 1), D(p), D(q)
    call("gkoc_cg_step_2_f64", ex.stream, rows, cols, dx.values, dx.ld, dr.values, dr.ld,
         dp.values, dp.ld, dq.values, dq.ld, dv(beta), dv(rho), dv(stop))
    ox, orr = oracle.cg_step_2(x, r, p, q, beta, rho, stop)
    assert np.array_equal(dx.to_numpy(), ox) and np.array_equal(dr.to_numpy(), orr)


def test_stop_kernels(gexec, oracle):
    import ctypes as C
    from ginkgo_amd._lib import call
    ex = gexec
    cols = 300
    rng = np.random.default_rng(8)
    tau = rng.uniform(0, 2, cols)
    orig = rng.uniform(0.5, 1.5, cols)
    stop = np.zeros(cols, np.uint8)
    stop[5] = 0x03
    for implicit in (False, True):
        for fin in (True, False):
            d_stop = ex.to_device(stop)
            flags = ex.zeros((2,), torch.uint8)
            allc, chg = C.c_int(0), C.c_int(0)
            name = "gkoc_implicit_residual_norm_f64" if implicit else "gkoc_residual_norm_f64"
            call(name, ex.stream, cols, ex.to_device(tau), ex.to_device(orig), C.c_double(0.9),
                 C.c_uint8(2), C.c_int(int(fin)), d_stop, flags, C.byref(allc), C.byref(chg))
            ra, rc, rs = oracle.residual_norm(tau, orig, 0.9, 2, fin, stop, implicit)
            assert (bool(allc.value), bool(chg.value)) == (ra, rc)
            assert np.array_equal(d_stop.cpu().numpy(), rs)
    # everything converges -> all_converged
    d_stop = ex.to_device(np.zeros(4, np.uint8))
    flags = ex.zeros((2,), torch.uint8)
    allc, chg = C.c_int(0), C.c_int(0)
    call("gkoc_residual_norm_f64", ex.stream, 4, ex.to_device(np.zeros(4)), ex.to_device(np.ones(4)),
         C.c_double(0.5), C.c_uint8(1), C.c_int(1), d_stop, flags, C.byref(allc), C.byref(chg))
    assert allc.value == 1 and chg.value == 1 and np.all(d_stop.cpu().numpy() == 0xC1)
    d_stop = ex.to_device(np.array([0, 0x81, 0], np.uint8))
    call("gkoc_set_all_statuses", ex.stream, 3, C.c_uint8(5), C.c_int(1), d_stop)
    assert d_stop.cpu().numpy().tolist() == [0x45, 0x81, 0x45]


# ------------------------------------------------------------------ Jacobi
def _block_matrix(seed, sizes, n_extra=3):
    """block-diagonal-dominant matrix whose diagonal blocks have identical row
    patterns (=> natural blocks) plus a few off-block entries"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    n = int(np.sum(sizes))
    a = sp.lil_matrix((n, n))
    s = 0
    for bs in sizes:
        blk = rng.uniform(-1, 1, (bs, bs)) + np.eye(bs) * bs
        a[s:s + bs, s:s + bs] = blk
        s += bs
    a = a.tocsr()
    a.sort_indices()
    return a


@pytest.mark.parametrize("max_bs", [1, 2, 3, 8, 13, 16, 32, 64])
def test_jacobi_blocks_generate_apply_bit_exact(gexec, oracle, max_bs):
    import ginkgo_amd as g
    rng = np.random.default_rng(max_bs)
    sizes = rng.integers(1, 9, 60)
    a = _block_matrix(max_bs, sizes)
    n = a.shape[0]
    rp, ci, v = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    da = g.Csr.from_arrays(gexec, (n, n), rp, ci, v)
    jac = g.Jacobi.build().with_max_block_size(max_bs).on(gexec).generate(da)
    b = rng.uniform(-1, 1, (n, 3))
    x0 = rng.uniform(-1, 1, (n, 3))
    if max_bs == 1:
        inv = oracle.jacobi_invert_diagonal(oracle.csr_extract_diagonal(n, n, rp, ci, v))
        assert np.array_equal(jac.inv_diag.cpu().numpy(), inv)
        x = g.Dense.create(gexec, (n, 3))
        jac.apply(g.Dense.from_numpy(gexec, b), x)
        assert np.array_equal(x.to_numpy(), oracle.jacobi_scalar_apply(inv, b))
        x = g.Dense.from_numpy(gexec, x0)
        jac.apply(g.scalar(gexec, 2.0), g.Dense.from_numpy(gexec, b), g.scalar(gexec, -1.0), x)
        assert np.array_equal(x.to_numpy(), oracle.jacobi_scalar_apply(inv, b, 2.0, -1.0, x0))
        return
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, max_bs)
    assert jac.num_blocks == nb
    assert np.array_equal(jac.block_pointers.cpu().numpy(), ptrs[:nb + 1])
    scheme = oracle.jacobi_storage_scheme(max_bs)
    assert (jac.scheme.block_offset, jac.scheme.group_offset, jac.scheme.group_power) == scheme
    blocks = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
    assert np.array_equal(jac.blocks.cpu().numpy(), blocks)
    x = g.Dense.create(gexec, (n, 3), stride=4)
    jac.apply(g.Dense.from_numpy(gexec, b, 5), x)
    assert np.array_equal(x.to_numpy(), oracle.jacobi_apply(nb, scheme, ptrs, blocks, b))
    x = g.Dense.from_numpy(gexec, x0)
    jac.apply(g.scalar(gexec, 2.0), g.Dense.from_numpy(gexec, b), g.scalar(gexec, -1.0), x)
    assert np.array_equal(x.to_numpy(), oracle.jacobi_apply(nb, scheme, ptrs, blocks, b, 2.0, -1.0, x0))


@pytest.mark.parametrize("ptr_kind", ["list", "int64", "int32", "tensor"])
def test_jacobi_with_block_pointers(gexec, oracle, ptr_kind):
    """Jacobi.with_block_pointers (jacobi.hpp:377-387): user-supplied blocks, given in any
    integer type, are converted to the matrix' index type; generate / apply == oracle"""
    import ginkgo_amd as g
    import torch
    rng = np.random.default_rng(5)
    sizes = rng.integers(1, 9, 40)
    a = _block_matrix(8, sizes)
    n = a.shape[0]
    rp, ci, v = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    # blocks different from the natural ones: pairs of rows, last block whatever is left
    ptrs = np.arange(0, n + 1, 2)
    if ptrs[-1] != n:
        ptrs = np.append(ptrs, n)
    given = {"list": [int(p) for p in ptrs], "int64": ptrs.astype(np.int64),
             "int32": ptrs.astype(np.int32), "tensor": torch.tensor(ptrs, dtype=torch.int64)}[ptr_kind]
    da = g.Csr.from_arrays(gexec, (n, n), rp, ci, v)
    jac = g.Jacobi.build().with_max_block_size(8).with_block_pointers(given).on(gexec).generate(da)
    nb = len(ptrs) - 1
    assert jac.num_blocks == nb and jac.block_pointers.dtype == torch.int32
    scheme = oracle.jacobi_storage_scheme(8)
    p32 = ptrs.astype(np.int32)
    blocks = oracle.jacobi_generate(rp, ci, v, nb, scheme, p32)
    assert np.array_equal(jac.blocks.cpu().numpy(), blocks)
    b = rng.uniform(-1, 1, (n, 2))
    x = g.Dense.create(gexec, (n, 2))
    jac.apply(g.Dense.from_numpy(gexec, b), x)
    assert np.array_equal(x.to_numpy(), oracle.jacobi_apply(nb, scheme, p32, blocks, b))
    for bad in ([1, n], [0, n - 1], [0, 3, 2, n], [0, 9, n]):
        with pytest.raises(g.GkoError):
            g.Jacobi.build().with_max_block_size(8).with_block_pointers(bad).on(gexec).generate(da)


def test_jacobi_pivoting_and_unsorted(gexec, oracle):
    """blocks that need row pivoting (zero / small diagonal) and an unsorted
    input matrix (Jacobi sorts a copy, jacobi.cpp:331-336)"""
    import ginkgo_amd as g
    import scipy.sparse as sp
    blk = np.array([[0., 2., 1., 0.], [4., 0., 0., 1.], [0., 0., 0., 3.], [1., 1., 5., 0.]])
    a = sp.block_diag([blk, blk.T + 1.0, np.array([[7.0]])]).tocsr()
    a.sort_indices()
    n = a.shape[0]
    rp, ci, v = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.copy()
    ci_u, v_u = ci.copy(), v.copy()
    for r in range(n):
        s, e = rp[r], rp[r + 1]
        ci_u[s:e], v_u[s:e] = ci_u[s:e][::-1], v_u[s:e][::-1]
    da = g.Csr.from_arrays(gexec, (n, n), rp, ci_u, v_u)
    assert not da.is_sorted_by_column_index()
    jac = g.Jacobi.build().with_max_block_size(4).on(gexec).generate(da)
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, 4)
    scheme = oracle.jacobi_storage_scheme(4)
    blocks = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
    assert jac.num_blocks == nb
    assert np.array_equal(jac.block_pointers.cpu().numpy(), ptrs[:nb + 1])
    assert np.array_equal(jac.blocks.cpu().numpy(), blocks)
    # the stored blocks really are the inverses
    b = np.arange(1.0, n + 1)
    x = g.Dense.create(gexec, (n, 1))
    jac.apply(g.Dense.from_numpy(gexec, b), x)
    assert np.allclose(a @ x.to_numpy()[:, 0], b, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("grid", [16, 40])
def test_jacobi_27pt(gexec, oracle, grid):
    """config 3's preconditioner: 27-pt Laplacian, max_block_size 8 => n/8 blocks
    of 8 consecutive rows"""
    import ginkgo_amd as g
    rp, ci, v = oracle.stencil_csr(3, grid)
    n = grid ** 3
    a = g.stencil_csr(gexec, 3, grid)
    jac = g.Jacobi.build().with_max_block_size(8).on(gexec).generate(a)
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, 8)
    assert jac.num_blocks == nb == n // 8
    assert np.array_equal(jac.block_pointers.cpu().numpy(), ptrs[:nb + 1])
    scheme = oracle.jacobi_storage_scheme(8)
    blocks = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
    assert np.array_equal(jac.blocks.cpu().numpy(), blocks)
    b = np.random.default_rng(1).uniform(-1, 1, n)
    x = g.Dense.create(gexec, (n, 1))
    jac.apply(g.Dense.from_numpy(gexec, b), x)
    assert np.array_equal(x.to_numpy()[:, 0], oracle.jacobi_apply(nb, scheme, ptrs, blocks, b))


# --------------------------------------------------------------- CG solves
def _solve(g, ex, a, b, x0, iters, red, precond_bs=None, baseline="rhs_norm"):
    f = (g.Cg.build().with_criteria(
        g.stop.Iteration.build().with_max_iters(iters),
        g.stop.ResidualNorm.build().with_reduction_factor(red).with_baseline(baseline)))
    if precond_bs:
        f = f.with_preconditioner(g.Jacobi.build().with_max_block_size(precond_bs))
    s = f.on(ex).generate(a)
    x = g.Dense.from_numpy(ex, x0)
    s.apply(g.Dense.from_numpy(ex, b), x)
    return s, x.to_numpy()[:, 0]


def test_cg_known_answers(gexec):
    import ginkgo_amd as g
    # reference/test/solver/cg_kernels.cpp:215-226
    a = g.Csr.from_arrays(gexec, (3, 3), np.array([0, 2, 5, 7], np.int32),
                          np.array([0, 1, 0, 1, 2, 1, 2], np.int32),
                          np.array([2., -1, -1, 2, -1, -1, 2]))
    s, x = _solve(g, gexec, a, np.array([-1., 3, 1]), np.zeros(3), 4, 1e-15)
    assert np.allclose(x, [1, 3, 2], rtol=1e-14)
    # :44-65 (mtx_big, cg_factory_big / big2), :407-462: dense 6x6 SPD systems,
    # ResidualNorm and ImplicitResidualNorm criteria, tolerance r<double>*1e2
    import scipy.sparse as sp
    m = np.array([[8828.0, 2673.0, 4150.0, -3139.5, 3829.5, 5856.0],
                  [2673.0, 10765.5, 1805.0, 73.0, 1966.0, 3919.5],
                  [4150.0, 1805.0, 6472.5, 2656.0, 2409.5, 3836.5],
                  [-3139.5, 73.0, 2656.0, 6048.0, 665.0, -132.0],
                  [3829.5, 1966.0, 2409.5, 665.0, 4240.5, 4373.5],
                  [5856.0, 3919.5, 3836.5, -132.0, 4373.5, 5678.0]])
    a = g.Csr.from_scipy(gexec, sp.csr_matrix(m))
    r_double = 10 * np.finfo(np.float64).eps      # core/test/utils.hpp:388-401
    b1 = np.array([1300083.0, 1018120.5, 906410.0, -42679.5, 846779.5, 1176858.5])
    s, x = _solve(g, gexec, a, b1, np.zeros(6), 100, r_double)
    assert rel_frobenius(x, [81.0, 55.0, 45.0, 5.0, 85.0, -10.0]) < r_double * 1e2
    b2 = np.array([886630.5, -172578.0, 684522.0, -65310.5, 455487.5, 607436.0])
    s, x = _solve(g, gexec, a, b2, np.zeros(6), 100, r_double)
    assert rel_frobenius(x, [33.0, -56.0, 81.0, -30.0, 21.0, 40.0]) < r_double * 1e2
    f = g.Cg.build().with_criteria(
        g.stop.Iteration.build().with_max_iters(100),
        g.stop.ImplicitResidualNorm.build().with_reduction_factor(r_double))
    xd = g.Dense.from_numpy(gexec, np.zeros(6))
    f.on(gexec).generate(a).apply(g.Dense.from_numpy(gexec, b2), xd)
    assert rel_frobenius(xd.to_numpy()[:, 0], [33.0, -56.0, 81.0, -30.0, 21.0, 40.0]) < r_double * 1e2


@pytest.mark.parametrize("case", ["5pt-256", "27pt-24", "27pt-40"])
@pytest.mark.parametrize("bs", [None, 1, 8])
def test_cg_vs_oracle(gexec, oracle, case, bs):
    """configs[0] (5-pt 2-D 256 x 256, CG + Jacobi) and reduced-size configs[2]
    (27-pt, CG + block-Jacobi(8), 1e-10): same iteration count (+-1), solution
    within 1e-9, true residual below the goal."""
    import ginkgo_amd as g
    nd, grid, restricted = {"5pt-256": (2, 256, True), "27pt-24": (3, 24, False),
                            "27pt-40": (3, 40, False)}[case]
    rp, ci, v = oracle.stencil_csr(nd, grid, restricted)
    n = grid ** nd
    a = g.stencil_csr(gexec, nd, grid, restricted)
    rhs = np.ones(n)
    s, x = _solve(g, gexec, a, rhs, np.zeros(n), 2000, 1e-10, bs)
    pre = {None: None, 1: "scalar", 8: "block"}[bs]
    xo, iters, rn = oracle.cg_solve(rp, ci, v, rhs, max_iters=2000, reduction=1e-10,
                                    precond=pre, max_block_size=8)
    assert s.has_converged
    assert abs(s.num_iterations - iters) <= 1
    assert rel_frobenius(x, xo) < 1e-9
    import scipy.sparse as sp
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    assert np.linalg.norm(rhs - A @ x) <= 1.01e-10 * np.linalg.norm(rhs)


def test_cg_iteration_limit_and_advanced_apply(gexec, oracle):
    import ginkgo_amd as g
    rp, ci, v = oracle.stencil_csr(3, 12)
    n = 12 ** 3
    a = g.stencil_csr(gexec, 3, 12)
    rhs = np.random.default_rng(5).uniform(-1, 1, n)
    s, x = _solve(g, gexec, a, rhs, np.zeros(n), 7, 1e-30)
    assert s.num_iterations == 7 and not s.has_converged
    xo, iters, _ = oracle.cg_solve(rp, ci, v, rhs, max_iters=7, reduction=1e-30)
    assert iters == 7 and rel_frobenius(x, xo) < 1e-12
    # initial_resnorm baseline with a non-zero guess
    x0 = np.full(n, 0.5)
    s, x = _solve(g, gexec, a, rhs, x0, 500, 1e-8, 8, baseline="initial_resnorm")
    xo, iters, _ = oracle.cg_solve(rp, ci, v, rhs, x0=x0, max_iters=500, reduction=1e-8,
                                   baseline="initial_resnorm", precond="block")
    assert abs(s.num_iterations - iters) <= 1 and rel_frobenius(x, xo) < 1e-7
    # x = beta*x + alpha*A^-1 b   (cg.cpp:184-200)
    xd = g.Dense.from_numpy(gexec, x0)
    s.apply(g.scalar(gexec, 2.0), g.Dense.from_numpy(gexec, rhs), g.scalar(gexec, -1.0), xd)
    assert rel_frobenius(xd.to_numpy()[:, 0], 2.0 * xo - x0) < 1e-6


def test_cg_multiple_rhs(gexec, oracle):
    """nrhs = 3 with per-column convergence (stop_status handling)"""
    import ginkgo_amd as g
    rp, ci, v = oracle.stencil_csr(2, 30, True)
    n = 900
    a = g.stencil_csr(gexec, 2, 30, True)
    rng = np.random.default_rng(3)
    B = np.stack([np.ones(n), rng.uniform(-1, 1, n), np.zeros(n)], axis=1)
    B[0, 2] = 1.0
    f = g.Cg.build().with_criteria(g.stop.Iteration.build().with_max_iters(500),
                                   g.stop.ResidualNorm.build().with_reduction_factor(1e-10))
    s = f.on(gexec).generate(a)
    X = g.Dense.from_numpy(gexec, np.zeros((n, 3)))
    s.apply(g.Dense.from_numpy(gexec, B), X)
    X = X.to_numpy()
    assert s.has_converged
    import scipy.sparse as sp
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    for j in range(3):
        assert np.linalg.norm(B[:, j] - A @ X[:, j]) <= 1.01e-10 * np.linalg.norm(B[:, j])


def test_stop_kernel_async_form(gexec, oracle):
    """NULL host results: flags stay on the device, nothing is synchronised"""
    from ginkgo_amd._lib import call, GkoError
    import ctypes as C
    ex = gexec
    cols = 300
    rng = np.random.default_rng(8)
    tau, orig = rng.uniform(0, 2, cols), rng.uniform(0.5, 1.5, cols)
    stop = np.zeros(cols, np.uint8)
    d_stop = ex.to_device(stop)
    flags = ex.zeros((2,), torch.uint8)
    call("gkoc_residual_norm_f64", ex.stream, cols, ex.to_device(tau), ex.to_device(orig),
         C.c_double(0.9), C.c_uint8(2), C.c_int(1), d_stop, flags, None, None)
    ra, rc, rs = oracle.residual_norm(tau, orig, 0.9, 2, True, stop, False)
    assert flags.cpu().numpy().tolist() == [int(ra), int(rc)]
    assert np.array_equal(d_stop.cpu().numpy(), rs)
    allc = C.c_int(0)
    with pytest.raises(GkoError):   # one host pointer without the other
        call("gkoc_residual_norm_f64", ex.stream, cols, ex.to_device(tau), ex.to_device(orig),
             C.c_double(0.9), C.c_uint8(2), C.c_int(1), d_stop, flags, C.byref(allc), None)


@pytest.mark.parametrize("bs", [None, 8])
def test_cg_deferred_check_changes_nothing(gexec, bs):
    """reading the criterion `check_lag` iterations late (the step kernels are
    masked by stop_status) gives bit-identical x, iteration count and status"""
    import ginkgo_amd as g
    grid = 24
    n = grid ** 3
    a = g.stencil_csr(gexec, 3, grid)
    rhs = np.random.default_rng(11).uniform(-1, 1, (n, 2))
    res = []
    for lag in (0, 1, 4, 7):
        f = (g.Cg.build().with_check_lag(lag).with_criteria(
            g.stop.ResidualNorm.build().with_reduction_factor(1e-9),
            g.stop.Iteration.build().with_max_iters(300)))
        if bs:
            f = f.with_preconditioner(g.Jacobi.build().with_max_block_size(bs))
        s = f.on(gexec).generate(a)
        x = g.Dense.from_numpy(gexec, np.zeros((n, 2)))
        s.apply(g.Dense.from_numpy(gexec, rhs), x)
        res.append((s.num_iterations, s.has_converged, x.to_numpy(), s.stop_status.cpu().numpy()))
    assert res[0][1]
    for r in res[1:]:
        assert r[0] == res[0][0] and r[1] == res[0][1]
        assert np.array_equal(r[2], res[0][2]) and np.array_equal(r[3], res[0][3])
    # iteration limit hit while checks are still pending
    for lag in (0, 4):
        f = (g.Cg.build().with_check_lag(lag).with_criteria(
            g.stop.Iteration.build().with_max_iters(5),
            g.stop.ResidualNorm.build().with_reduction_factor(1e-30)))
        s = f.on(gexec).generate(a)
        x = g.Dense.from_numpy(gexec, np.zeros((n, 2)))
        s.apply(g.Dense.from_numpy(gexec, rhs), x)
        assert s.num_iterations == 5 and not s.has_converged


# ------------------------------------------------- fused extensions (gkoc_x_*)
def _xwork(ex, n, dtype=torch.float64):
    import ctypes as C
    from ginkgo_amd._lib import lib
    nbytes = lib().gkoc_x_workspace_bytes(C.c_int64(n), C.c_size_t(8))
    return ex.alloc(((nbytes + 7) // 8,), dtype), nbytes


@pytest.mark.parametrize("grid", [5, 17, 40])
def test_fused_spmv_dot(gexec, grid):
    """c bit-identical to csr::spmv, dot = <b, c> within the reduction tolerance"""
    import ctypes as C
    import ginkgo_amd as g
    from ginkgo_amd._lib import call
    n = grid ** 3
    a = g.stencil_csr(gexec, 3, grid)
    bv = np.random.default_rng(grid).uniform(-1, 1, n)
    b = g.Dense.from_numpy(gexec, bv)
    c0, c1 = g.Dense.create(gexec, (n, 1)), g.Dense.create(gexec, (n, 1))
    a.apply(b, c0)
    work, nbytes = _xwork(gexec, n)
    out = g.Dense.create(gexec, (1, 1))
    a.apply_dot(b, c1, out, work)
    assert np.array_equal(c0.to_numpy(), c1.to_numpy())
    ref = float(np.dot(bv, c0.to_numpy()[:, 0]))
    assert abs(out.to_numpy()[0, 0] - ref) <= 1e-13 * np.sum(np.abs(bv * c0.to_numpy()[:, 0]))
    # deterministic
    out2 = g.Dense.create(gexec, (1, 1))
    a.apply_dot(b, c1, out2, work)
    assert out.to_numpy()[0, 0] == out2.to_numpy()[0, 0]
    # workspace check
    from ginkgo_amd._lib import GkoError
    with pytest.raises(GkoError):
        call("gkoc_x_csr_spmv_dot_f64_i32", gexec.stream, n, a.row_ptrs, a.col_idxs, a.values,
             b.values, c1.values, out.values, work, C.c_size_t(nbytes - 8))


@pytest.mark.parametrize("max_bs", [2, 4, 8, 16])
def test_fused_jacobi_apply_dot(gexec, max_bs):
    import ginkgo_amd as g
    grid = 24
    n = grid ** 3
    a = g.stencil_csr(gexec, 3, grid)
    m = g.Jacobi.build().with_max_block_size(max_bs).on(gexec).generate(a)
    rv = np.random.default_rng(max_bs).uniform(-1, 1, n)
    r = g.Dense.from_numpy(gexec, rv)
    z0, z1 = g.Dense.create(gexec, (n, 1)), g.Dense.create(gexec, (n, 1))
    m.apply(r, z0)
    assert m.can_fuse_dot(r)
    work, _ = _xwork(gexec, n)
    out = g.Dense.create(gexec, (1, 1))
    m.apply_dot(r, z1, out, work)
    assert np.array_equal(z0.to_numpy(), z1.to_numpy())
    zz = z0.to_numpy()[:, 0]
    assert abs(out.to_numpy()[0, 0] - float(np.dot(rv, zz))) <= 1e-13 * np.sum(np.abs(rv * zz))
    # layouts outside the fast path are refused, the caller keeps the two-kernel form
    m13 = g.Jacobi.build().with_max_block_size(13).on(gexec).generate(a)
    assert not m13.can_fuse_dot(r)


@pytest.mark.parametrize("n", [1, 777, 100003, 1 << 20])
def test_fused_step_2_norm(gexec, oracle, n):
    import ctypes as C
    from ginkgo_amd._lib import call
    import ginkgo_amd as g
    rng = np.random.default_rng(n)
    x, r, p, q = (rng.uniform(-1, 1, n) for _ in range(4))
    work, nbytes = _xwork(gexec, n)
    for beta, stopped in ((0.37, 0), (0.0, 0), (0.37, 0x81)):
        dx, dr = gexec.to_device(x), gexec.to_device(r)
        stop = gexec.to_device(np.array([stopped], np.uint8))
        out = g.Dense.create(gexec, (1, 1))
        call("gkoc_x_cg_step_2_norm_f64", gexec.stream, n, dx, dr, gexec.to_device(p),
             gexec.to_device(q), gexec.to_device(np.array([beta])), gexec.to_device(np.array([0.81])),
             stop, out.values, C.c_int(1), work, C.c_size_t(nbytes))
        ox, orr = oracle.cg_step_2(x.copy(), r.copy(), p.copy(), q.copy(), np.array([beta]),
                                   np.array([0.81]), np.array([stopped], np.uint8))
        assert np.array_equal(dx.cpu().numpy(), ox.reshape(-1))
        assert np.array_equal(dr.cpu().numpy(), orr.reshape(-1))
        nr = np.linalg.norm(orr)
        assert abs(out.to_numpy()[0, 0] - nr) <= 1e-13 * max(nr, 1e-300)


@pytest.mark.parametrize("bs", [None, 8, 13])
def test_cg_fused_vs_unfused(gexec, bs):
    """same iteration count and the same solution to rounding with and without
    the fused kernels (vectors are bit-identical, scalars differ by the tree)"""
    import ginkgo_amd as g
    grid = 24
    n = grid ** 3
    a = g.stencil_csr(gexec, 3, grid)
    rhs = np.random.default_rng(21).uniform(-1, 1, n)
    res = []
    for fused in (False, True):
        f = (g.Cg.build().with_fused_kernels(fused).with_criteria(
            g.stop.Iteration.build().with_max_iters(300),
            g.stop.ResidualNorm.build().with_reduction_factor(1e-10)))
        if bs:
            f = f.with_preconditioner(g.Jacobi.build().with_max_block_size(bs))
        s = f.on(gexec).generate(a)
        x = g.Dense.from_numpy(gexec, np.zeros(n))
        s.apply(g.Dense.from_numpy(gexec, rhs), x)
        res.append((s.num_iterations, s.has_converged, x.to_numpy()[:, 0], s.residual_norm))
    assert res[0][1] and res[1][1]
    assert abs(res[0][0] - res[1][0]) <= 1
    assert rel_frobenius(res[1][2], res[0][2]) < 1e-9


@pytest.mark.parametrize("bs", [None, 8])
def test_cg_hip_graph_changes_nothing(gexec, bs):
    """two iterations captured in a hipGraph and replayed: bit-identical x,
    iteration count and stop status vs the eager loop; reusable across applies"""
    import ginkgo_amd as g
    grid = 20
    n = grid ** 3
    a = g.stencil_csr(gexec, 3, grid)
    rhs = np.random.default_rng(4).uniform(-1, 1, n)
    res = {}
    for mode in (False, True):
        for max_it in (1000, 7, 8):
            f = (g.Cg.build().with_hip_graph(mode).with_criteria(
                g.stop.Iteration.build().with_max_iters(max_it),
                g.stop.ResidualNorm.build().with_reduction_factor(1e-10)))
            if bs:
                f = f.with_preconditioner(g.Jacobi.build().with_max_block_size(bs))
            s = f.on(gexec).generate(a)
            x = g.Dense.from_numpy(gexec, np.zeros(n))
            for _ in range(2):                       # second apply reuses the captured graph
                x.fill(0.0)
                s.apply(g.Dense.from_numpy(gexec, rhs), x)
            res[(mode, max_it)] = (s.num_iterations, s.has_converged, x.to_numpy(),
                                   s.stop_status.cpu().numpy())
    for max_it in (1000, 7, 8):
        e, h = res[(False, max_it)], res[(True, max_it)]
        assert e[0] == h[0] and e[1] == h[1]
        assert np.array_equal(e[2], h[2]) and np.array_equal(e[3], h[3])
    assert res[(True, 1000)][1] and res[(True, 7)][0] == 7 and not res[(True, 7)][1]


@pytest.mark.parametrize("graph", [False, True])
def test_cg_float32_fused_paths(gexec, graph):
    """the f32 instantiations of the fused kernels / hipGraph loop: CG + block-Jacobi(8)
    in single precision reaches the single-precision residual level and agrees
    with the double-precision solution to ~1e-4"""
    import ginkgo_amd as g
    grid = 16
    n = grid ** 3
    rhs = np.random.default_rng(2).uniform(-1, 1, n)
    sols = {}
    for dt, tol in ((torch.float64, 1e-10), (torch.float32, 1e-5)):
        a = g.stencil_csr(gexec, 3, grid, dtype=dt)
        npdt = np.float64 if dt == torch.float64 else np.float32
        s = (g.Cg.build().with_hip_graph(graph)
             .with_criteria(g.stop.Iteration.build().with_max_iters(500),
                            g.stop.ResidualNorm.build().with_reduction_factor(tol))
             .with_preconditioner(g.Jacobi.build().with_max_block_size(8))
             .on(gexec).generate(a))
        x = g.Dense.from_numpy(gexec, np.zeros(n, npdt))
        s.apply(g.Dense.from_numpy(gexec, rhs.astype(npdt)), x)
        assert s.has_converged
        sols[dt] = x.to_numpy()[:, 0].astype(np.float64)
    assert rel_frobenius(sols[torch.float32], sols[torch.float64]) < 2e-4


# ------------------------------------------------ block-Jacobi, reduced storage precision
@pytest.mark.parametrize("bs", [2, 4, 8, 16])
def test_jacobi_reduced_storage_bit_exact(gexec, oracle, bs):
    """Jacobi::storage_optimization = precision_reduction(p, n) for all blocks: the stored
    blocks (raw bytes of the used part) and simple / advanced apply against the oracle,
    which is pinned to the reference (tests/golden/jacobi_storage.npz) - bit-exact"""
    import ginkgo_amd as g
    rng = np.random.default_rng(bs)
    rp, ci, v = oracle.stencil_csr(3, 9)
    v = v * rng.uniform(0.05, 20.0, len(v))
    n = len(rp) - 1
    a = g.Csr.from_arrays(gexec, (n, n), rp, ci, v)
    b, x0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, bs)
    scheme = oracle.jacobi_storage_scheme(bs)
    full = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
    for p_, n_ in ((0, 1), (0, 2), (1, 0), (1, 1), (2, 0)):
        prec = (p_ << 4) | n_
        m = g.Jacobi.build().with_max_block_size(bs).with_storage_optimization(p_, n_).on(gexec).generate(a)
        st = oracle.jacobi_convert_storage(nb, scheme, full, prec)
        # the narrow entries of a group occupy its first bytes
        width = {0x01: 4, 0x02: 2, 0x10: 4, 0x11: 2, 0x20: 2}[prec]
        go = scheme[1]
        dev = m.blocks.cpu().numpy().view(np.uint8).reshape(-1, go * 8)[:, :go * width]
        ref = st.view(np.uint8).reshape(-1, go * 8)[:, :go * width]
        mask = _block_byte_mask(scheme, ptrs[:nb + 1], width)
        assert np.array_equal(dev[mask], ref[mask]), (bs, hex(prec))
        x = g.Dense.create(gexec, (n, 1))
        m.apply(g.Dense.from_numpy(gexec, b), x)
        assert np.array_equal(x.to_numpy()[:, 0], oracle.jacobi_apply_stored(nb, scheme, ptrs, st, prec, b))
        x = g.Dense.from_numpy(gexec, x0)
        m.apply(g.scalar(gexec, 0.7), g.Dense.from_numpy(gexec, b), g.scalar(gexec, -1.1), x)
        assert np.array_equal(x.to_numpy()[:, 0],
                              oracle.jacobi_apply_stored(nb, scheme, ptrs, st, prec, b, 0.7, -1.1, x0))


def _block_byte_mask(scheme, ptrs, width):
    """bytes of the group-major storage (first go*width bytes of every group) that belong to
    an entry of a block"""
    bo, go, gp = scheme
    stride = bo << gp
    nb = len(ptrs) - 1
    groups = (nb + (1 << gp) - 1) >> gp
    mask = np.zeros((groups, go * width), dtype=bool)
    for blk in range(nb):
        bsz = int(ptrs[blk + 1] - ptrs[blk])
        g_, off = blk >> gp, bo * (blk & ((1 << gp) - 1))
        for c in range(bsz):
            for r in range(bsz):
                e = off + r + c * stride
                mask[g_, e * width:(e + 1) * width] = True
    return mask


def test_jacobi_reduced_storage_golden_and_cg(gexec, oracle):
    """the reference's own apply results; CG with a float- and a half-stored block-Jacobi
    converges to the same solution"""
    import os
    import ginkgo_amd as g
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "jacobi_storage.npz"))
    rp, ci, v, b, x0 = (gold[k] for k in ("row_ptrs", "cols", "vals", "b", "x0"))
    n = len(rp) - 1
    a = g.Csr.from_arrays(gexec, (n, n), rp, ci, v)
    for bs in (4, 8, 16):
        for p_, n_ in ((0, 1), (0, 2), (1, 0), (1, 1), (2, 0)):
            m = g.Jacobi.build().with_max_block_size(bs).with_storage_optimization(p_, n_).on(gexec).generate(a)
            x = g.Dense.create(gexec, (n, 1))
            m.apply(g.Dense.from_numpy(gexec, b), x)
            assert np.array_equal(x.to_numpy()[:, 0], gold[f"bs{bs}_p{p_}n{n_}_apply"])
            x = g.Dense.from_numpy(gexec, x0)
            m.apply(g.scalar(gexec, 0.7), g.Dense.from_numpy(gexec, b), g.scalar(gexec, -1.1), x)
            assert np.array_equal(x.to_numpy()[:, 0], gold[f"bs{bs}_p{p_}n{n_}_apply_adv"])
    grid = 20
    a = g.stencil_csr(gexec, 3, grid)
    rp, ci, v = oracle.stencil_csr(3, grid)
    rhs = np.ones(grid ** 3)
    sols = []
    for prec in (None, (0, 1), (0, 2)):
        pf = g.Jacobi.build().with_max_block_size(8)
        if prec:
            pf = pf.with_storage_optimization(*prec)
        s = (g.Cg.build().with_criteria(g.stop.Iteration.build().with_max_iters(300),
                                        g.stop.ResidualNorm.build().with_reduction_factor(1e-10))
             .with_preconditioner(pf).on(gexec).generate(a))
        x = g.Dense.from_numpy(gexec, np.zeros(grid ** 3))
        s.apply(g.Dense.from_numpy(gexec, rhs), x)
        assert s.has_converged
        r = rhs - oracle.csr_spmv(rp, ci, v, x.to_numpy()[:, 0])
        assert np.linalg.norm(r) <= 2e-10 * np.linalg.norm(rhs)
        sols.append((x.to_numpy()[:, 0], s.num_iterations))
    assert abs(sols[1][1] - sols[0][1]) <= 2 and abs(sols[2][1] - sols[0][1]) <= 3
    with pytest.raises(g.NotSupported):      # 13 x 13 blocks are not stored in 64-wide groups
        g.Jacobi.build().with_max_block_size(13).with_storage_optimization(0, 1).on(gexec).generate(a)


@pytest.mark.parametrize("bs", [2, 4, 8, 16, 5, 13, 32])
def test_jacobi_adaptive_precision_bit_exact(gexec, oracle, bs):
    """storage_optimization autodetect and block-wise requests: the chosen precision per
    block, the condition numbers, the stored bytes and both applies against the oracle
    (pinned to the reference in tests/test_oracle_cpu.py) - all bit-exact; block sizes that are
    not a power of two (block_offset 5, 13: what Ginkgo's own jacobi tests use) as well"""
    import ginkgo_amd as g
    from adaptive_cases import graded_block_matrix
    rp, ci, v = graded_block_matrix(24, bs, bs)
    n = len(rp) - 1
    a = g.Csr.from_arrays(gexec, (n, n), rp, ci, v)
    rng = np.random.default_rng(bs)
    b, x0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, bs)
    scheme = oracle.jacobi_storage_scheme(bs)
    seen = set()
    for acc, req in ((1e-1, None), (1e-3, None), (1e-2, [0x01, 0xff, 0x20, 0x00, 0x11, 0xff, 0x02, 0x10]),
                     (1e-1, [0x20] * 3 + [0xff] * 5)):
        f = g.Jacobi.build().with_max_block_size(bs).with_accuracy(acc)
        if req is None:
            f = f.with_storage_optimization("autodetect")
        else:
            f = f.with_storage_optimization([("autodetect" if q == 0xff else (q >> 4, q & 15)) for q in req])
        m = f.on(gexec).generate(a)
        assert m.num_blocks == nb
        blocks_o, prec_o, cond_o = oracle.jacobi_generate_adaptive(rp, ci, v, nb, scheme, ptrs[:nb + 1], acc, req)
        prec_d = m.precisions.cpu().numpy()
        assert np.array_equal(prec_d, prec_o), (bs, acc)
        assert np.array_equal(m.conditioning.cpu().numpy(), cond_o), (bs, acc)
        seen |= set(int(p) for p in prec_o)
        # stored bytes, group by group, in the width of the group's type
        go, gp = scheme[1], scheme[2]
        dev = m.blocks.cpu().numpy().view(np.uint8).reshape(-1, go * 8)
        ref = blocks_o.view(np.uint8).reshape(-1, go * 8)
        for grp in range(dev.shape[0]):
            width = {0x00: 8, 0x01: 4, 0x02: 2, 0x10: 4, 0x11: 2, 0x20: 2}[int(prec_o[grp << gp])]
            sub_ptrs = ptrs[grp << gp:min(((grp + 1) << gp), nb) + 1]
            mask = _block_byte_mask(scheme, sub_ptrs, width)[0]
            assert np.array_equal(dev[grp, :go * width][mask], ref[grp, :go * width][mask]), (bs, acc, grp)
        x = g.Dense.create(gexec, (n, 1))
        m.apply(g.Dense.from_numpy(gexec, b), x)
        assert np.array_equal(x.to_numpy()[:, 0],
                              oracle.jacobi_apply_adaptive(nb, scheme, ptrs[:nb + 1], blocks_o, prec_o, b))
        x = g.Dense.from_numpy(gexec, x0)
        m.apply(g.scalar(gexec, 0.7), g.Dense.from_numpy(gexec, b), g.scalar(gexec, -1.1), x)
        assert np.array_equal(x.to_numpy()[:, 0],
                              oracle.jacobi_apply_adaptive(nb, scheme, ptrs[:nb + 1], blocks_o, prec_o, b, 0.7, -1.1, x0))
    assert {0x00, 0x01, 0x02} <= seen and (len(seen) >= 4 or bs >= 13)


def test_jacobi_adaptive_cg(gexec, oracle):
    """CG with the autodetected storage: same solution, within a few iterations"""
    import ginkgo_amd as g
    grid = 20
    a = g.stencil_csr(gexec, 3, grid)
    rp, ci, v = oracle.stencil_csr(3, grid)
    rhs = np.ones(grid ** 3)
    its = []
    for adaptive in (False, True):
        pf = g.Jacobi.build().with_max_block_size(8)
        if adaptive:
            pf = pf.with_storage_optimization("autodetect")
        s = (g.Cg.build().with_criteria(g.stop.Iteration.build().with_max_iters(300),
                                        g.stop.ResidualNorm.build().with_reduction_factor(1e-10))
             .with_preconditioner(pf).on(gexec).generate(a))
        x = g.Dense.from_numpy(gexec, np.zeros(grid ** 3))
        s.apply(g.Dense.from_numpy(gexec, rhs), x)
        r = rhs - oracle.csr_spmv(rp, ci, v, x.to_numpy()[:, 0])
        assert s.has_converged and np.linalg.norm(r) <= 2e-10 * np.linalg.norm(rhs)
        its.append(s.num_iterations)
        if adaptive:     # well-conditioned 8 x 8 blocks: half storage everywhere
            assert set(s.preconditioner.precisions.cpu().numpy().tolist()) == {0x02}
    assert abs(its[0] - its[1]) <= 3


@pytest.mark.parametrize("itype", [np.int32, np.int64])
@pytest.mark.parametrize("max_bs", [2, 4, 8, 16])
def test_fused_step_2_jacobi_apply(gexec, oracle, max_bs, itype):
    """gkoc_x_cg_step_2_jacobi_apply: x, r as cg::step_2 (oracle), z as jacobi::simple_apply of the
    new r (oracle), bit for bit; <r,z> with the bits of the unfused apply + dot; ||r||; a zero beta
    and a stopped column leave x and r alone and still produce z"""
    import ctypes as C
    from ginkgo_amd._lib import call
    import ginkgo_amd as g
    rng = np.random.default_rng(max_bs)
    sizes = rng.integers(1, max_bs + 1, 3000)
    a = _block_matrix(max_bs, sizes)
    n = a.shape[0]
    rp, ci, v = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    jac = g.Jacobi.build().with_max_block_size(max_bs).on(gexec).generate(
        g.Csr.from_arrays(gexec, (n, n), rp.astype(itype), ci.astype(itype), v))
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, max_bs)
    scheme = oracle.jacobi_storage_scheme(max_bs)
    blocks = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
    x0, r0, p, q = (rng.uniform(-1, 1, n) for _ in range(4))
    work, nbytes = _xwork(gexec, n)
    assert jac.can_fuse_step_2(g.Dense.from_numpy(gexec, r0))
    for beta, stopped in ((0.37, 0), (0.0, 0), (0.37, 0x81)):
        x, r = g.Dense.from_numpy(gexec, x0), g.Dense.from_numpy(gexec, r0)
        z = g.Dense.from_numpy(gexec, np.full(n, np.nan))
        rho_out, nrm = g.Dense.create(gexec, (1, 1)), g.Dense.create(gexec, (1, 1))
        stop = gexec.to_device(np.array([stopped], np.uint8))
        jac.step_2_apply_dot(x, r, g.Dense.from_numpy(gexec, p), g.Dense.from_numpy(gexec, q),
                             g.scalar(gexec, beta), g.scalar(gexec, 0.81), stop, z, rho_out, nrm, True, work)
        ox, orr = oracle.cg_step_2(x0.copy(), r0.copy(), p.copy(), q.copy(), np.array([beta]),
                                   np.array([0.81]), np.array([stopped], np.uint8))
        assert np.array_equal(x.to_numpy()[:, 0], ox.reshape(-1))
        assert np.array_equal(r.to_numpy()[:, 0], orr.reshape(-1))
        oz = oracle.jacobi_apply(nb, scheme, ptrs, blocks, orr.reshape(-1, 1))[:, 0]
        assert np.array_equal(z.to_numpy()[:, 0], oz)
        # the unfused pair on the same r: identical <r,z> bits, ||r|| to the tree's tolerance
        z2, rho2 = g.Dense.create(gexec, (n, 1)), g.Dense.create(gexec, (1, 1))
        jac.apply_dot(r, z2, rho2, work)
        assert np.array_equal(rho_out.to_numpy(), rho2.to_numpy())
        nr = np.linalg.norm(orr)
        assert abs(nrm.to_numpy()[0, 0] - nr) <= 1e-13 * nr


@pytest.mark.parametrize("bs", [4, 8])
def test_cg_fused_step_2_apply_changes_nothing(gexec, bs):
    """CG with step_2 and the next preconditioner application in one kernel against the
    two-kernel sequence: x bit-identical, same iteration count and stop status"""
    import ginkgo_amd as g
    grid = 22
    n = grid ** 3
    a = g.stencil_csr(gexec, 3, grid)
    rhs = np.random.default_rng(8).uniform(-1, 1, n)
    res = {}
    for fused in (False, True):
        for max_it in (1000, 9):
            s = (g.Cg.build().with_fused_step_2_apply(fused).with_hip_graph(False).with_criteria(
                g.stop.Iteration.build().with_max_iters(max_it),
                g.stop.ResidualNorm.build().with_reduction_factor(1e-10))
                .with_preconditioner(g.Jacobi.build().with_max_block_size(bs)).on(gexec).generate(a))
            x = g.Dense.from_numpy(gexec, np.zeros(n))
            s.apply(g.Dense.from_numpy(gexec, rhs), x)
            res[(fused, max_it)] = (s.num_iterations, s.has_converged, x.to_numpy(),
                                    s.stop_status.cpu().numpy())
    for max_it in (1000, 9):
        e, f = res[(False, max_it)], res[(True, max_it)]
        assert e[0] == f[0] and e[1] == f[1]
        assert np.array_equal(e[2], f[2]) and np.array_equal(e[3], f[3])
    assert res[(True, 1000)][1] and res[(True, 9)][0] == 9
