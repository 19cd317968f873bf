"""Pins the oracle (oracle/gko_oracle.c) before anything is compared with it:

 1. known-answer vectors of the reference's own unit tests (cited inline);
 2. golden fixtures under tests/golden/*.npz, produced by the UNMODIFIED
    reference (tests/golden/make_golden.py) - these also run on the GPU box;
 3. when oracle/_ref is built (this container), the live reference on further
    random inputs, bit-for-bit.
"""
import os

import numpy as np
import pytest

from util import random_csr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


# ------------------------------------------------- 1. known-answer vectors
def test_csr_known_answers(oracle):
    # reference/test/matrix/csr_kernels.cpp:84-108 fixture [[1,3,2],[0,5,0]]
    rp = np.array([0, 3, 4], np.int32)
    ci = np.array([0, 1, 2, 1], np.int32)
    v = np.array([1., 3., 2., 5.])
    x = np.array([2., 1., 4.])
    assert oracle.csr_spmv(rp, ci, v, x).tolist() == [13.0, 5.0]               # :353-364
    assert oracle.csr_spmv(rp, ci, v, x, -1.0, 2.0, np.array([1., 2.])).tolist() == [-11.0, -1.0]  # :505-518
    nan = np.array([np.nan, np.nan])
    assert oracle.csr_spmv(rp, ci, v, x, -1.0, 0.0, nan).tolist() == [-13.0, -5.0]   # :521-534
    # :414-430 multiple right-hand sides
    xm = np.array([[2., 3.], [1., -1.5], [4., 2.5]])
    assert oracle.csr_spmv(rp, ci, v, xm).tolist() == [[13.0, 3.5], [5.0, -7.5]]
    # ELL / SELL-P of the same matrix (reference/test/matrix/ell_kernels.cpp:78-88,
    # sellp_kernels.cpp:56-66) give the same products
    k, stride, ec, ev = oracle.csr_to_ell(rp, ci, v)
    assert (k, stride) == (3, 2)
    assert oracle.ell_spmv(2, k, stride, ec, ev, x).tolist() == [13.0, 5.0]
    sets, lens, sc, sv = oracle.csr_to_sellp(rp, ci, v, slice_size=64)
    assert lens.tolist() == [3] and sets.tolist() == [0, 3]
    assert oracle.sellp_spmv(2, 64, sets, lens, sc, sv, x).tolist() == [13.0, 5.0]


def test_cg_kernel_known_answers(oracle):
    # reference/test/solver/cg_kernels.cpp:113-213 (initialize / step_1 / step_2
    # incl. the division-by-zero conventions)
    b = np.full((2, 2), 2.0)
    r, z, p, q, prev_rho, rho, stop = oracle.cg_initialize(b)
    assert np.all(r == 2) and np.all(z == 0) and np.all(p == 0) and np.all(q == 0)
    assert np.all(prev_rho == 1) and np.all(rho == 0) and np.all(stop == 0)
    # :140-160 step_1: p = z + rho/prev_rho * p ; prev_rho == 0 => p = z
    p = np.array([[1., 2.], [3., 4.]])
    z = np.array([[1., 1.], [1., 1.]])
    out = oracle.cg_step_1(p, z, np.array([2., 3.]), np.array([8., 0.]), np.zeros(2, np.uint8))
    assert out.tolist() == [[1.25, 1.0], [1.75, 1.0]]
    # :176-213 step_2: beta == 0 => untouched
    x, r = oracle.cg_step_2(np.ones((2, 2)), np.ones((2, 2)), p, np.full((2, 2), 2.0),
                            np.array([8., 0.]), np.array([2., 3.]), np.zeros(2, np.uint8))
    assert x.tolist() == [[1.25, 1.0], [1.75, 1.0]] and r.tolist() == [[0.5, 1.0], [0.5, 1.0]]
    # a stopped column is never touched
    out = oracle.cg_step_1(p, z, np.array([2., 3.]), np.array([8., 1.]), np.array([0, 0x81], np.uint8))
    assert out[:, 1].tolist() == [2.0, 4.0]


def test_cg_solver_known_answers(oracle):
    # reference/test/solver/cg_kernels.cpp:215-226
    rp = np.array([0, 2, 5, 7], np.int32)
    ci = np.array([0, 1, 0, 1, 2, 1, 2], np.int32)
    v = np.array([2., -1, -1, 2, -1, -1, 2])
    x, it, _ = oracle.cg_solve(rp, ci, v, np.array([-1., 3, 1]), max_iters=4, reduction=1e-15)
    assert np.allclose(x, [1, 3, 2], rtol=1e-14)
    # :44-65, :407-446 dense 6 x 6 SPD systems, tolerance r<double> * 1e2
    import scipy.sparse as sp
    m = sp.csr_matrix(np.array([[8828.0, 2673.0, 4150.0, -3139.5, 3829.5, 5856.0],
                                [2673.0, 10765.5, 1805.0, 73.0, 1966.0, 3919.5],
                                [4150.0, 1805.0, 6472.5, 2656.0, 2409.5, 3836.5],
                                [-3139.5, 73.0, 2656.0, 6048.0, 665.0, -132.0],
                                [3829.5, 1966.0, 2409.5, 665.0, 4240.5, 4373.5],
                                [5856.0, 3919.5, 3836.5, -132.0, 4373.5, 5678.0]]))
    rp, ci, v = m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data
    r_double = 10 * np.finfo(np.float64).eps
    for b, sol in [([1300083.0, 1018120.5, 906410.0, -42679.5, 846779.5, 1176858.5],
                    [81.0, 55.0, 45.0, 5.0, 85.0, -10.0]),
                   ([886630.5, -172578.0, 684522.0, -65310.5, 455487.5, 607436.0],
                    [33.0, -56.0, 81.0, -30.0, 21.0, 40.0])]:
        x, it, _ = oracle.cg_solve(rp, ci, v, np.array(b), max_iters=100, reduction=r_double)
        err = np.linalg.norm(x - sol) / np.linalg.norm(sol)
        assert err < r_double * 1e2


def test_jacobi_known_answers(oracle):
    # reference/test/preconditioner/jacobi_kernels.cpp:129-260 style: natural
    # blocks {2,1,2} agglomerated with max_block_size 3 -> [0,3,5]
    import scipy.sparse as sp
    a = sp.block_diag([np.array([[4., 1.], [2., 4.]]), np.array([[3.]]),
                       np.array([[4., -2.], [-1., 4.]])]).tocsr()
    a.sort_indices()
    rp, ci, v = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, 3)
    assert nb == 2 and ptrs[:3].tolist() == [0, 3, 5]
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, 2)
    assert nb == 3 and ptrs[:4].tolist() == [0, 2, 3, 5]
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, 8)
    assert nb == 1 and ptrs[:2].tolist() == [0, 5]
    # inversion: [[4,-2],[-1,4]]^-1 = 1/14 [[4,2],[1,4]]
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, 2)
    scheme = oracle.jacobi_storage_scheme(2)
    assert scheme == (2, 128, 5)      # jacobi.hpp:589-627 with stride 64
    blocks = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
    bo, go, gp = scheme
    stride = bo << gp
    blk = blocks[2 * bo:]              # third block of group 0
    inv = np.array([[blk[0], blk[stride]], [blk[1], blk[1 + stride]]])
    assert np.allclose(inv, np.array([[4., 2.], [1., 4.]]) / 14.0, rtol=1e-15)
    x = oracle.jacobi_apply(nb, scheme, ptrs, blocks, np.array([5., 6., 3., 2., 3.]))
    assert np.allclose(a @ x, [5., 6., 3., 2., 3.], rtol=1e-14)
    # scalar Jacobi: zero diagonal entries are replaced by one (:582-590)
    assert oracle.jacobi_invert_diagonal(np.array([2., 0., -4.])).tolist() == [0.5, 1.0, -0.25]


def test_stop_known_answers(oracle):
    # reference/test/stop/residual_norm_kernels.cpp + stopping_status.hpp:85-121
    allc, chg, st = oracle.residual_norm(np.array([1e-3, 2.0]), np.array([1.0, 1.0]), 1e-2, 3, True,
                                         np.zeros(2, np.uint8))
    assert (allc, chg) == (False, True) and st.tolist() == [0xC3, 0]
    allc, chg, st = oracle.residual_norm(np.array([1e-3, 1e-5]), np.array([1.0, 1.0]), 1e-2, 3, False,
                                         np.array([0, 0x45], np.uint8))
    assert (allc, chg) == (True, True) and st.tolist() == [0x83, 0x45]
    allc, chg, st = oracle.residual_norm(np.array([4e-4]), np.array([1.0]), 1e-2, 1, True,
                                         np.zeros(1, np.uint8), implicit=True)
    assert not allc            # sqrt(4e-4) = 2e-2 > 1e-2


def test_dense_known_answers(oracle):
    # reference/test/matrix/dense_kernels.cpp ComputesDot / ComputesNorm2 / Scales
    x = np.array([[1., 0.], [2., 3.], [2., 4.]])
    assert oracle.dense_dot(x, x).tolist() == [9.0, 25.0]
    assert oracle.dense_norm2(x).tolist() == [3.0, 5.0]
    assert oracle.dense_scale([2.0], x).tolist() == [[2., 0.], [4., 6.], [4., 8.]]
    assert oracle.dense_scale([0.0], np.full((2, 2), np.nan)).tolist() == [[0., 0.], [0., 0.]]
    assert oracle.dense_add_scaled([2.0, -1.0], x, np.ones((3, 2))).tolist() == [[3., 1.], [5., -2.], [5., -3.]]


# ---------------------------------------------------- 2. golden fixtures
def test_golden_spmv(oracle):
    g = gold("spmv_532x231.npz")
    rp, ci, v, b, c0 = g["row_ptrs"], g["cols"], g["vals"], g["b"], g["c0"]
    assert np.array_equal(oracle.csr_spmv(rp, ci, v, b), g["y"])
    assert np.array_equal(oracle.csr_spmv(rp, ci, v, b, 2.0, -1.0, c0), g["y_adv"])
    k, stride, ec, ev = oracle.csr_to_ell(rp, ci, v)
    assert (k, stride) == (int(g["ell_k"]), int(g["ell_stride"]))
    assert np.array_equal(ec, g["ell_cols"]) and np.array_equal(ev, g["ell_vals"])
    assert np.array_equal(oracle.ell_spmv(532, k, stride, ec, ev, b), g["y_ell"])
    sets, lens, sc, sv = oracle.csr_to_sellp(rp, ci, v, 64, 1)
    assert np.array_equal(sets, g["slice_sets"]) and np.array_equal(lens, g["slice_lengths"])
    # rows past num_rows in the last slice are uninitialised in the reference
    live = np.ones(len(sc), bool)
    last = int(sets[-2]) * 64
    for i in range(int(lens[-1])):
        live[last + i * 64 + (532 - 512):last + (i + 1) * 64] = False
    assert np.array_equal(sc[live], g["sellp_cols"][live])
    assert np.array_equal(sv[live], g["sellp_vals"][live])
    assert np.array_equal(oracle.sellp_spmv(532, 64, sets, lens, sc, sv, b), g["y_sellp"])


def test_golden_stencil(oracle):
    g = gold("stencil.npz")
    names = sorted({k[:-5] for k in g.files if k.endswith("_meta")})
    assert len(names) == 7
    for name in names:
        meta = g[name + "_meta"]
        nd, dims, pos, tls, restricted, ls = int(meta[0]), meta[1:4], meta[4:7], int(meta[7]), int(meta[8]), int(meta[9])
        nsub = int(np.prod(dims[:nd]))
        grid = int(round((tls * nsub) ** (1.0 / nd)))
        r, c, v, ls2 = oracle.stencil_subdomain(nd, dims[:nd], pos[:nd], grid, restricted)
        assert ls2 == ls
        assert np.array_equal(r, g[name + "_rows"]), name
        assert np.array_equal(c, g[name + "_cols"]), name
        assert np.array_equal(v, g[name + "_vals"]), name
    # the direct CSR generator agrees with COO -> CSR of the subdomain generator
    r, c, v, _ = oracle.stencil_subdomain(3, [1, 1, 1], [0, 0, 0], 6, False)
    rp64, ci64, vv = oracle.coo_to_csr(r, c, v, 0, 216)
    rp, ci, v2 = oracle.stencil_csr(3, 6)
    assert np.array_equal(rp, rp64) and np.array_equal(ci, ci64) and np.array_equal(v2, vv)


def test_golden_krylov(oracle):
    g = gold("krylov_27pt_10.npz")
    rp, ci, v, rhs = g["row_ptrs"], g["cols"], g["vals"], g["rhs"]
    for bs in (4, 8, 13, 32):
        nb, ptrs = oracle.jacobi_find_blocks(rp, ci, bs)
        assert np.array_equal(ptrs[:nb + 1], g[f"jac{bs}_ptrs"])
        scheme = oracle.jacobi_storage_scheme(bs)
        assert scheme == tuple(int(s) for s in g[f"jac{bs}_scheme"])
        blocks = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
        ref_blocks = g[f"jac{bs}_blocks"]
        # padding entries of the interleaved storage are uninitialised in the
        # reference: compare the slots that belong to a block
        mask = _block_mask(scheme, ptrs[:nb + 1], len(ref_blocks))
        assert np.array_equal(blocks[mask], ref_blocks[mask])
        assert np.array_equal(oracle.jacobi_apply(nb, scheme, ptrs, blocks, rhs), g[f"jac{bs}_apply"])
        assert np.array_equal(oracle.jacobi_apply(nb, scheme, ptrs, blocks, rhs, 2.0, -1.0, np.ones(1000)),
                              g[f"jac{bs}_apply_adv"])
    for bs, pre in [(0, None), (1, "scalar"), (4, "block"), (8, "block"), (13, "block"), (32, "block")]:
        x, it, rn = oracle.cg_solve(rp, ci, v, np.ones(1000), precond=pre, max_block_size=max(bs, 1))
        assert it == int(g[f"cg{bs}_iters"][0]), bs
        assert np.array_equal(x, g[f"cg{bs}_x"]), bs
        assert rn == float(g[f"cg{bs}_resnorm"][0])
    x, it, _ = oracle.cg_solve(rp, ci, v, rhs, x0=np.full(1000, 0.5), max_iters=9, reduction=1e-30,
                               baseline="initial_resnorm", precond="block")
    assert it == int(g["cg_lim_iters"][0]) == 9 and np.array_equal(x, g["cg_lim_x"])


def test_golden_gmres(oracle):
    """core/solver/gmres.cpp:321-621 driver + gmres / common_gmres kernels: the
    oracle reproduces the reference's iterate, iteration count and Givens
    residual-norm estimate bit-for-bit for MGS, CGS and CGS2, with and without
    restarts / block-Jacobi."""
    g = gold("krylov_27pt_10.npz")
    rp, ci, v, rhs = g["row_ptrs"], g["cols"], g["vals"], g["rhs"]
    for ortho in ("mgs", "cgs", "cgs2"):
        for kd, bs in ((100, 0), (7, 8)):
            x, it, rn = oracle.gmres_solve(rp, ci, v, rhs, krylov_dim=kd, ortho=ortho, max_iters=300,
                                           reduction=1e-9, precond="block" if bs else None,
                                           max_block_size=8)
            key = f"gmres_{ortho}_{kd}_{bs}"
            assert it == int(g[key + "_iters"][0]), key
            assert rn == float(g[key + "_resnorm"][0]), key
            assert np.array_equal(x, g[key + "_x"]), key


def _block_mask(scheme, ptrs, size):
    bo, go, gp = scheme
    stride = bo << gp
    mask = np.zeros(size, bool)
    for b in range(len(ptrs) - 1):
        bs = int(ptrs[b + 1] - ptrs[b])
        base = go * (b >> gp) + bo * (b & ((1 << gp) - 1))
        for c in range(bs):
            mask[base + c * stride:base + c * stride + bs] = True
    return mask


def test_golden_jacobi_blocks(oracle):
    g = gold("jacobi_blocks.npz")
    rp, ci, v, b = g["row_ptrs"], g["cols"], g["vals"], g["b"]
    for bs in (2, 6, 16):
        nb, ptrs = oracle.jacobi_find_blocks(rp, ci, bs)
        assert np.array_equal(ptrs[:nb + 1], g[f"jac{bs}_ptrs"])
        scheme = tuple(int(s) for s in g[f"jac{bs}_scheme"])
        assert scheme == oracle.jacobi_storage_scheme(bs)
        blocks = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
        mask = _block_mask(scheme, ptrs[:nb + 1], len(blocks))
        assert np.array_equal(blocks[mask], g[f"jac{bs}_blocks"][mask])
        assert np.array_equal(oracle.jacobi_apply(nb, scheme, ptrs, blocks, b), g[f"jac{bs}_apply"])


def test_golden_dense(oracle):
    g = gold("dense.npz")
    assert np.array_equal(oracle.dense_dot(g["x"], g["y"]), g["dot"])
    assert np.array_equal(oracle.dense_norm2(g["x"]), g["norm2"])


# ------------------------------------------------ 3. live reference (if built)
def _ref():
    from oracle import ref_shim
    if not ref_shim.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return ref_shim


@pytest.mark.parametrize("grid", [1, 2, 3, 7, 16])
def test_cpu_baseline_generator_is_the_benchmark_matrix(oracle, grid):
    """bench.py's cpu_baseline lets the OpenMP threads generate the 27-pt matrix straight into the
    OmpExecutor's arrays (parallel first touch, oracle/ref_shim.cpp): the same row pointers, column
    indices and values as the oracle's / the reference's generator"""
    import ctypes as C
    rl = _ref().lib()
    rl.ref_stencil27_nnz.restype = C.c_int64
    rl.ref_stencil27_csr.restype = C.c_int64
    rp, ci, v = oracle.stencil_csr(3, grid)
    nnz = rl.ref_stencil27_nnz(C.c_int64(grid))
    assert nnz == len(v)
    rp2, ci2, v2 = np.zeros(grid ** 3 + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
    assert rl.ref_stencil27_csr(C.c_int64(grid), rp2.ctypes.data_as(C.c_void_p),
                                ci2.ctypes.data_as(C.c_void_p), v2.ctypes.data_as(C.c_void_p)) == nnz
    assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2) and np.array_equal(v, v2)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_live_reference_spmv(oracle, seed):
    ref = _ref()
    rng = np.random.default_rng(seed)
    rows, cols = int(rng.integers(50, 400)), int(rng.integers(50, 400))
    rp, ci, v = random_csr(rows, cols, 0.07, seed, unsorted=bool(seed % 2), empty_rows=(1,))
    b = rng.uniform(-1, 1, (cols, 2))
    c0 = rng.uniform(-1, 1, (rows, 2))
    h = ref.CsrHandle("reference", rp, ci, v, n_cols=cols)
    assert np.array_equal(oracle.csr_spmv(rp, ci, v, b), h.spmv(b))
    assert np.array_equal(oracle.csr_spmv(rp, ci, v, b, 0.3, 1.7, c0), h.spmv(b, 0.3, 1.7, c0))
    assert np.array_equal(oracle.csr_spmv(rp, ci, v, b, 0.3, 0.0, c0), h.spmv(b, 0.3, 0.0, c0))
    # OmpExecutor agrees with the sequential reference to rounding
    ho = ref.CsrHandle("omp", rp, ci, v, n_cols=cols)
    assert np.allclose(ho.spmv(b), h.spmv(b), rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("case", [(2, 40, True, 1), (2, 24, False, 4), (3, 9, False, 8), (3, 8, True, 3)])
def test_live_reference_cg_and_jacobi(oracle, case):
    ref = _ref()
    nd, grid, restricted, bs = case
    rp, ci, v = oracle.stencil_csr(nd, grid, restricted)
    n = grid ** nd
    h = ref.CsrHandle("reference", rp, ci, v)
    rhs = np.random.default_rng(bs).uniform(-1, 1, n)
    pre = "scalar" if bs == 1 else "block"
    xo, ito, rno = oracle.cg_solve(rp, ci, v, rhs, max_iters=300, reduction=1e-9, precond=pre, max_block_size=bs)
    xr, itr, rnr = h.cg_solve(rhs, max_iters=300, reduction=1e-9, precond_block_size=bs)
    assert ito == itr and rno == rnr and np.array_equal(xo, xr)
    if bs > 1:
        nb, scheme, ptrs, blocks = h.jacobi_generate(bs)
        nbo, po = oracle.jacobi_find_blocks(rp, ci, bs)
        assert nbo == nb and np.array_equal(po[:nb + 1], ptrs)
        bo = oracle.jacobi_generate(rp, ci, v, nb, scheme, po)
        mask = _block_mask(scheme, ptrs, len(blocks))
        assert np.array_equal(bo[mask], blocks[mask])


@pytest.mark.parametrize("ortho", ["mgs", "cgs", "cgs2"])
def test_live_reference_gmres(oracle, ortho):
    ref = _ref()
    rp, ci, v = oracle.stencil_csr(2, 20, False)
    h = ref.CsrHandle("reference", rp, ci, v)
    rhs = np.random.default_rng(11).uniform(-1, 1, 400)
    for kd, bs in ((100, 0), (6, 4), (3, 1)):
        xr, itr, rnr = h.gmres_solve(rhs, krylov_dim=kd, ortho=ortho, max_iters=150, reduction=1e-8,
                                     precond_block_size=bs)
        pre = None if bs == 0 else ("scalar" if bs == 1 else "block")
        xo, ito, rno = oracle.gmres_solve(rp, ci, v, rhs, krylov_dim=kd, ortho=ortho, max_iters=150,
                                          reduction=1e-8, precond=pre, max_block_size=max(bs, 1))
        assert (ito, rno) == (itr, rnr) and np.array_equal(xo, xr)


def test_live_reference_stencil_subdomains(oracle):
    ref = _ref()
    for nd, dims, grid, restricted in [(3, [2, 1, 1], 8, False), (3, [3, 1, 1], 7, False),
                                       (2, [4, 1], 12, True), (3, [2, 2, 1], 6, True)]:
        nsub = int(np.prod(dims))
        for rank in range(nsub):
            pos = [rank % dims[0], (rank // dims[0]) % dims[1]] + ([rank // (dims[0] * dims[1])] if nd == 3 else [])
            tls = grid ** nd // nsub
            assert round((tls * nsub) ** (1 / nd)) == grid
            a = oracle.stencil_subdomain(nd, dims, pos, grid, restricted)
            b = ref.stencil_subdomain(nd, dims, pos, tls, restricted)
            assert a[3] == b[3]
            for u, w in zip(a[:3], b[:3]):
                assert np.array_equal(u, w)


# ----------------------------------------------------------------------------
# Bicgstab / Cgs / Fcg / PipeCg
def test_krylov_family_known_answers(oracle):
    import krylov_family_cases as kc
    for solver, kernel, inp, stop, exp in kc.CASES:
        for dt in (np.float64, np.float32):
            arr = kc.materialise(solver, kernel, inp, stop, dt)
            oracle.krylov_step(f"{solver}_{kernel}", 2, 2, *arr.values())
            kc.check(arr, exp)


def _family_cases(g):
    for mname, kinds in (("sym", ("bicgstab", "cgs", "fcg", "pipe_cg")), ("nonsym", ("bicgstab", "cgs"))):
        mat = tuple(g[f"{mname}_{k}"] for k in ("row_ptrs", "cols", "vals"))
        for kind in kinds:
            yield mname, mat, g[f"{mname}_rhs"], kind


def test_golden_krylov_family(oracle):
    g = gold("krylov_family.npz")
    for mname, (rp, ci, v), rhs, kind in _family_cases(g):
        n = len(rp) - 1
        for bs, pre in ((0, None), (1, "scalar"), (8, "block")):
            x, it, rn = oracle.krylov_solve(kind, rp, ci, v, rhs, max_iters=400, reduction=1e-9,
                                            precond=pre, max_block_size=max(bs, 1))
            it_ref, rn_ref = g[f"{mname}_{kind}_{bs}_it_rn"]
            assert (it, rn) == (int(it_ref), float(rn_ref)), (mname, kind, bs)
            assert np.array_equal(x, g[f"{mname}_{kind}_{bs}_x"]), (mname, kind, bs)
        x, it, rn = oracle.krylov_solve(kind, rp, ci, v, rhs, x0=np.full(n, 0.5), max_iters=6,
                                        reduction=1e-30, baseline="initial_resnorm", precond="block")
        it_ref, rn_ref = g[f"{mname}_{kind}_lim_it_rn"]
        assert (it, rn) == (int(it_ref), float(rn_ref)) and it == 6
        assert np.array_equal(x, g[f"{mname}_{kind}_lim_x"])


@pytest.mark.parametrize("kind", ["bicgstab", "cgs", "fcg", "pipe_cg"])
def test_live_reference_krylov_family(oracle, kind):
    ref = _ref()
    rp, ci, v = oracle.stencil_csr(3, 7)
    if kind in ("bicgstab", "cgs"):                    # also a non-symmetric operator
        rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
        v = v.copy()
        v[ci > rows] *= 0.7
    n = len(rp) - 1
    h = ref.CsrHandle("reference", rp, ci, v)
    rhs = np.random.default_rng(5).uniform(-1, 1, n)
    for bs in (0, 1, 4):
        pre = None if bs == 0 else ("scalar" if bs == 1 else "block")
        xo, ito, rno = oracle.krylov_solve(kind, rp, ci, v, rhs, max_iters=200, reduction=1e-10,
                                           precond=pre, max_block_size=max(bs, 1))
        xr, itr, rnr = h.krylov_solve(kind, rhs, max_iters=200, reduction=1e-10, precond_block_size=bs)
        assert (ito, rno) == (itr, rnr) and np.array_equal(xo, xr)


# ----------------------------------------------------------------------------
# Coo / Hybrid
def test_coo_known_answers(oracle):
    """reference/test/matrix/coo_kernels.cpp: the 2 x 3 matrix [[1,3,2],[0,5,0]]
    (AppliesToDenseVector :311-320, AppliesLinearCombinationToDenseVector :406-417,
    AppliesAddToDenseVector :509-518, AppliesLinearCombinationAddToDenseVector :557-567)"""
    rows = np.array([0, 0, 0, 1], np.int32)
    cols = np.array([0, 1, 2, 1], np.int32)
    vals = np.array([1.0, 3.0, 2.0, 5.0])
    x = np.array([2.0, 1.0, 4.0])
    assert np.array_equal(oracle.coo_apply("spmv", 2, rows, cols, vals, x), [13.0, 5.0])
    assert np.array_equal(oracle.coo_apply("advanced_spmv", 2, rows, cols, vals, x,
                                           -1.0, 2.0, np.array([1.0, 2.0])), [-11.0, -1.0])
    assert np.array_equal(oracle.coo_apply("spmv2", 2, rows, cols, vals, x, c=np.array([2.0, 1.0])), [15.0, 6.0])
    assert np.array_equal(oracle.coo_apply("advanced_spmv2", 2, rows, cols, vals, x,
                                           -1.0, c=np.array([1.0, 2.0])), [-12.0, -3.0])


def test_golden_coo_hybrid(oracle):
    g = gold("coo_hybrid.npz")
    rp, rows, cols, vals, b, c0, perm = (g[k] for k in ("row_ptrs", "rows", "cols", "vals", "b", "c0", "perm"))
    for mode in ("spmv", "advanced_spmv", "spmv2", "advanced_spmv2"):
        assert np.array_equal(oracle.coo_apply(mode, 532, rows, cols, vals, b, 2.0, -1.0, c0), g[mode])
        assert np.array_equal(oracle.coo_apply(mode, 532, rows[perm], cols[perm], vals[perm], b, 2.0, -1.0, c0),
                              g[mode + "_shuffled"])
    # sorted COO = CSR row sums started from c (what the device path relies on)
    assert np.array_equal(g["spmv"], oracle.csr_spmv(rp, cols, vals, b))
    assert np.array_equal(g["spmv2"], oracle.csr_spmv(rp, cols, vals, b, alpha=1.0, beta=1.0, c=c0))
    assert np.array_equal(g["advanced_spmv2"], oracle.csr_spmv(rp, cols, vals, b, alpha=2.0, beta=1.0, c=c0))
    for lim in (0, 4, 9, 1000):
        k, st = (int(t) for t in g[f"hyb{lim}_shape"])
        ec, ev, crp, cr, cc, cv = oracle.csr_to_hybrid(rp, cols, vals, k, st)
        for got, name in ((ec, "ell_cols"), (ev, "ell_vals"), (cr, "coo_rows"), (cc, "coo_cols"), (cv, "coo_vals")):
            assert np.array_equal(got, g[f"hyb{lim}_{name}"]), (lim, name)
        y = oracle.ell_spmv(532, k, st, ec, ev, b) if k else np.zeros((532, 3))
        y = oracle.coo_apply("spmv2", 532, cr, cc, cv, b, c=y)
        assert np.array_equal(y, g[f"hyb{lim}_apply"])


def test_live_reference_coo_hybrid(oracle):
    ref = _ref()
    rng = np.random.default_rng(8)
    rp, ci, v = random_csr(97, 64, 0.15, seed=9)
    rows = np.repeat(np.arange(97, dtype=np.int32), np.diff(rp))
    b, c0 = rng.uniform(-1, 1, (64, 2)), rng.uniform(-1, 1, (97, 2))
    for mode in ("spmv", "advanced_spmv", "spmv2", "advanced_spmv2"):
        for p in (np.arange(len(v)), rng.permutation(len(v))):
            assert np.array_equal(oracle.coo_apply(mode, 97, rows[p], ci[p], v[p], b, 0.7, 1.3, c0),
                                  ref.coo_apply(mode, 97, 64, rows[p], ci[p], v[p], b, 0.7, 1.3, c0))
    h = ref.CsrHandle("reference", rp, ci, v, n_cols=64)
    for lim in (0, 2, 11):
        k, st, ec, ev, cr, cc, cv = h.to_hybrid(lim)
        got = oracle.csr_to_hybrid(rp, ci, v, k, st)
        for a_, b_ in zip((got[0], got[1], got[3], got[4], got[5]), (ec, ev, cr, cc, cv)):
            assert np.array_equal(a_, b_)


# ----------------------------------------------------------------------------
# Ir / Chebyshev
def test_chebyshev_kernel_known_answers(oracle):
    import ctypes as C
    import krylov_family_cases as kc
    for dt, suf in ((np.float64, "f64"), (np.float32, "f32")):
        for kernel in ("init_update", "update"):
            inner, upd, out = (np.array(m, dtype=dt) for m in (kc.CHEB_INNER, kc.CHEB_UPDATE, kc.CHEB_OUTPUT))
            f = getattr(oracle.lib(), f"oracle_chebyshev_{kernel}_{suf}")
            coeffs = [C.c_double(0.5)] + ([C.c_double(0.25)] if kernel == "update" else [])
            f(C.c_int64(3), C.c_int64(3), C.c_int64(3), *coeffs, oracle._p(inner), oracle._p(upd),
              oracle._p(out))
            kc.check_chebyshev(kernel, inner, upd, out)


def test_golden_stationary(oracle):
    g = gold("stationary.npz")
    rp, ci, v, rhs = g["row_ptrs"], g["cols"], g["vals"], g["rhs"]
    n = len(rp) - 1
    for bs, pre in ((0, None), (1, "scalar"), (8, "block")):
        it_ref, rn_ref, relax = g[f"ir_{bs}_it_rn"]
        x, it, rn = oracle.krylov_solve("ir", rp, ci, v, rhs, max_iters=400, reduction=1e-6, precond=pre,
                                        max_block_size=max(bs, 1), relaxation=float(relax))
        assert (it, rn) == (int(it_ref), float(rn_ref)) and np.array_equal(x, g[f"ir_{bs}_x"])
        it_ref, rn_ref, f0, f1 = g[f"chebyshev_{bs}_it_rn"]
        x, it, rn = oracle.krylov_solve("chebyshev", rp, ci, v, rhs, x0=np.full(n, 0.1), max_iters=400,
                                        reduction=1e-6, precond=pre, max_block_size=max(bs, 1),
                                        foci=(float(f0), float(f1)))
        assert (it, rn) == (int(it_ref), float(rn_ref)) and np.array_equal(x, g[f"chebyshev_{bs}_x"])
    for kind, kw in (("ir", dict(relaxation=0.9)), ("chebyshev", dict(foci=(0.02, 2.0)))):
        x, it, rn = oracle.krylov_solve(kind, rp, ci, v, rhs, x0=np.full(n, 0.5), max_iters=7, reduction=1e-30,
                                        baseline="initial_resnorm", precond="block", **kw)
        it_ref, rn_ref = g[f"{kind}_lim_it_rn"]
        assert (it, rn) == (int(it_ref), float(rn_ref)) and np.array_equal(x, g[f"{kind}_lim_x"])


def test_live_reference_stationary(oracle):
    ref = _ref()
    rp, ci, v = oracle.stencil_csr(2, 14, True)
    n = len(rp) - 1
    h = ref.CsrHandle("reference", rp, ci, v)
    rhs = np.random.default_rng(6).uniform(-1, 1, n)
    for bs, pre in ((0, None), (1, "scalar"), (4, "block")):
        relax, foci = (0.2, (0.05, 8.0)) if bs == 0 else (0.8, (0.02, 2.0))
        xo, ito, rno = oracle.krylov_solve("ir", rp, ci, v, rhs, max_iters=300, reduction=1e-5, precond=pre,
                                           max_block_size=max(bs, 1), relaxation=relax)
        xr, itr, rnr = h.stationary_solve("ir", rhs, max_iters=300, reduction=1e-5, precond_block_size=bs,
                                          relaxation=relax)
        assert (ito, rno) == (itr, rnr) and np.array_equal(xo, xr)
        xo, ito, rno = oracle.krylov_solve("chebyshev", rp, ci, v, rhs, max_iters=300, reduction=1e-5,
                                           precond=pre, max_block_size=max(bs, 1), foci=foci)
        xr, itr, rnr = h.stationary_solve("chebyshev", rhs, max_iters=300, reduction=1e-5,
                                          precond_block_size=bs, foci=foci)
        assert (ito, rno) == (itr, rnr) and np.array_equal(xo, xr)


def test_live_reference_transpose(oracle):
    ref = _ref()
    for nr, nc, d, seed in ((60, 45, 0.2, 3), (5, 300, 0.1, 4), (200, 3, 0.5, 5), (40, 40, 0.0, 6)):
        rp, ci, v = random_csr(nr, nc, d, seed=seed)
        h = ref.CsrHandle("reference", rp, ci, v, n_cols=nc)
        for got, want in zip(oracle.csr_transpose(nr, nc, rp, ci, v), h.transpose()):
            assert np.array_equal(got, want)


def test_transpose_known_answer(oracle):
    """reference/test/matrix/csr_kernels.cpp:1612-1632 (SquareMtxIsTransposable)"""
    rp = np.array([0, 3, 4, 6], np.int32)
    ci = np.array([0, 1, 2, 1, 1, 2], np.int32)
    v = np.array([1.0, 3.0, 2.0, 5.0, 1.5, 2.0])
    trp, tc, tv = oracle.csr_transpose(3, 3, rp, ci, v)
    dense = np.zeros((3, 3))
    for r in range(3):
        dense[r, tc[trp[r]:trp[r + 1]]] = tv[trp[r]:trp[r + 1]]
    assert np.array_equal(dense, [[1.0, 0.0, 0.0], [3.0, 5.0, 1.5], [2.0, 0.0, 2.0]])


# ----------------------------------------------------------------------------
# block-Jacobi with reduced storage precision (fixed storage_optimization)
PRECS = ((0, 1), (0, 2), (1, 0), (1, 1), (2, 0))


def test_reduced_storage_conversions(oracle):
    """half: round to nearest even, exponents below the normal range give signed zero,
    above give infinity (include/ginkgo/core/base/half.hpp:405-433); truncated types keep
    the upper bits (core/base/extended_float.hpp:72-89)"""
    scheme = (1, 64, 6)                 # one 1 x 1 block per lane: 64 values = one group
    vals = np.array([1.0, -1.0, 0.1, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 65504.0, 65520.0, 1e5, 6e-5,
                     6.2e-5, -5e-8, 3.14159265358979, 2.0 ** -14, 1e-300, -0.0] + [0.0] * 49)
    def roundtrip(prec):
        st = oracle.jacobi_convert_storage(64, scheme, vals, prec)
        ptrs = np.arange(65, dtype=np.int32)
        return oracle.jacobi_apply_stored(64, scheme, ptrs, st, prec, np.ones(64))
    h = roundtrip(0x02)
    assert h[0] == 1.0 and h[1] == -1.0
    assert h[2] == float(np.float16(np.float32(0.1)))
    assert h[3] == 1.0                           # tie -> even
    assert h[4] == 1.0 + 2.0 ** -9               # tie -> even (upwards)
    assert h[5] == 65504.0 and np.isinf(h[6]) and np.isinf(h[7])
    assert h[8] == 0.0 and h[9] == float(np.float16(6.2e-5))      # 6e-5 < 2^-14: flushed
    assert h[10] == 0.0                          # -5e-8: flushed (the sign is lost in 0 + -0)
    assert h[12] == 2.0 ** -14 and h[13] == 0.0
    f = roundtrip(0x01)
    assert np.array_equal(f[:15], vals[:15].astype(np.float32).astype(np.float64))
    t32 = roundtrip(0x10)
    want = (vals.view(np.uint64) & np.uint64(0xFFFFFFFF00000000)).view(np.float64)
    assert np.array_equal(t32, want)
    t16 = roundtrip(0x20)
    assert np.array_equal(t16, (vals.view(np.uint64) & np.uint64(0xFFFF000000000000)).view(np.float64))
    b16 = roundtrip(0x11)
    wf = (vals.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    assert np.array_equal(b16, wf.astype(np.float64))


def test_golden_jacobi_storage(oracle):
    g = gold("jacobi_storage.npz")
    rp, ci, v, b, x0 = (g[k] for k in ("row_ptrs", "cols", "vals", "b", "x0"))
    for bs in (4, 8, 16):
        nb, ptrs = oracle.jacobi_find_blocks(rp, ci, bs)
        scheme = oracle.jacobi_storage_scheme(bs)
        full = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
        for p_, n_ in PRECS:
            prec = (p_ << 4) | n_
            st = oracle.jacobi_convert_storage(nb, scheme, full, prec)
            key = f"bs{bs}_p{p_}n{n_}"
            assert np.array_equal(oracle.jacobi_apply_stored(nb, scheme, ptrs, st, prec, b), g[key + "_apply"]), key
            assert np.array_equal(oracle.jacobi_apply_stored(nb, scheme, ptrs, st, prec, b, 0.7, -1.1, x0),
                                  g[key + "_apply_adv"]), key


def test_live_reference_jacobi_storage(oracle):
    ref = _ref()
    rng = np.random.default_rng(4)
    rp, ci, v = oracle.stencil_csr(2, 17, True)
    v = v * rng.uniform(0.1, 10.0, len(v))
    n = len(rp) - 1
    h = ref.CsrHandle("reference", rp, ci, v)
    b = rng.uniform(-1, 1, n)
    for bs in (3, 8):
        for p_, n_ in PRECS:
            nb, scheme, ptrs, _ = h.jacobi_generate_prec(bs, p_, n_)
            nbo, po = oracle.jacobi_find_blocks(rp, ci, bs)
            full = oracle.jacobi_generate(rp, ci, v, nb, scheme, po)
            st = oracle.jacobi_convert_storage(nb, scheme, full, (p_ << 4) | n_)
            assert np.array_equal(oracle.jacobi_apply_stored(nb, scheme, po, st, (p_ << 4) | n_, b), h.jacobi_apply(b))


@pytest.mark.parametrize("bs", [2, 4, 8, 16])
def test_live_reference_jacobi_adaptive(oracle, bs):
    """autodetect and block-wise storage_optimization: chosen precisions, condition numbers
    and apply results of the oracle restatement equal the reference's, bit for bit"""
    ref = _ref()
    from adaptive_cases import graded_block_matrix
    rp, ci, v = graded_block_matrix(24, bs, bs)
    n = len(rp) - 1
    h = ref.CsrHandle("reference", rp, ci, v)
    b = np.random.default_rng(1).uniform(-1, 1, n)
    seen = set()
    for acc, req in ((1e-1, None), (1e-3, None), (1e-2, [0x01, 0xff, 0x20, 0x00, 0x11, 0xff, 0x02, 0x10]),
                     (1e-1, [0x20] * 3 + [0xff] * 5)):
        nb, scheme, ptrs, _, prec_r, cond_r = h.jacobi_generate_adaptive(bs, acc, req)
        y_r = h.jacobi_apply(b)
        blocks_o, prec_o, cond_o = oracle.jacobi_generate_adaptive(rp, ci, v, nb, scheme, ptrs, acc, req)
        assert np.array_equal(prec_r, prec_o) and np.array_equal(cond_r, cond_o)
        assert np.array_equal(oracle.jacobi_apply_adaptive(nb, scheme, ptrs, blocks_o, prec_o, b), y_r)
        seen |= set(int(p) for p in prec_o)
    assert {0x00, 0x01, 0x02} <= seen and (len(seen) >= 4 or bs == 16)


# ----------------------------------------------------------------------------
# Bicg - needs csr::conj_transpose and Jacobi::conj_transpose
def test_golden_bicg(oracle):
    g = gold("bicg.npz")
    rp, ci, v, rhs = g["row_ptrs"], g["cols"], g["vals"], g["rhs"]
    n = len(rp) - 1
    for got, want in zip(oracle.csr_transpose(n, n, rp, ci, v), (g["t_row_ptrs"], g["t_cols"], g["t_vals"])):
        assert np.array_equal(got, want)
    for bs, pre in ((0, None), (1, "scalar"), (8, "block")):
        x, it, rn = oracle.krylov_solve("bicg", rp, ci, v, rhs, max_iters=400, reduction=1e-9, precond=pre,
                                        max_block_size=max(bs, 1))
        it_ref, rn_ref = g[f"bicg_{bs}_it_rn"]
        assert (it, rn) == (int(it_ref), float(rn_ref)) and np.array_equal(x, g[f"bicg_{bs}_x"])
    x, it, rn = oracle.krylov_solve("bicg", rp, ci, v, rhs, x0=np.full(n, 0.5), max_iters=6, reduction=1e-30,
                                    baseline="initial_resnorm", precond="block")
    assert (it, rn) == tuple(g["bicg_lim_it_rn"]) and np.array_equal(x, g["bicg_lim_x"])


def test_live_reference_bicg(oracle):
    ref = _ref()
    rp, ci, v = oracle.stencil_csr(3, 7)
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    v = v.copy()
    v[ci > rows] *= 0.7
    h = ref.CsrHandle("reference", rp, ci, v)
    rhs = np.random.default_rng(5).uniform(-1, 1, len(rp) - 1)
    for bs in (0, 1, 4, 8):
        pre = None if bs == 0 else ("scalar" if bs == 1 else "block")
        xo, ito, rno = oracle.krylov_solve("bicg", rp, ci, v, rhs, max_iters=200, reduction=1e-10, precond=pre,
                                           max_block_size=max(bs, 1))
        xr, itr, rnr = h.krylov_solve("bicg", rhs, max_iters=200, reduction=1e-10, precond_block_size=bs)
        assert (ito, rno) == (itr, rnr) and np.array_equal(xo, xr)


def test_golden_and_live_gcr(oracle):
    g = gold("bicg.npz")
    rp, ci, v, rhs = g["row_ptrs"], g["cols"], g["vals"], g["rhs"]
    for kd in (100, 6):
        for bs, pre in ((0, None), (8, "block")):
            x, it, rn = oracle.krylov_solve("gcr", rp, ci, v, rhs, max_iters=400, reduction=1e-9, precond=pre,
                                            max_block_size=max(bs, 1), krylov_dim=kd)
            assert (it, rn) == tuple(g[f"gcr_{kd}_{bs}_it_rn"]) and np.array_equal(x, g[f"gcr_{kd}_{bs}_x"])
    if os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libgko_ref_shim.so")):
        ref = _ref()
        rp, ci, v = oracle.stencil_csr(3, 6)
        h = ref.CsrHandle("reference", rp, ci, v)
        b = np.random.default_rng(3).uniform(-1, 1, len(rp) - 1)
        for kd, bs in ((100, 1), (4, 4)):
            pre = "scalar" if bs == 1 else "block"
            xo, ito, rno = oracle.krylov_solve("gcr", rp, ci, v, b, max_iters=200, reduction=1e-10, precond=pre,
                                               max_block_size=bs, krylov_dim=kd)
            xr, itr, rnr = h.gcr_solve(b, krylov_dim=kd, max_iters=200, reduction=1e-10, precond_block_size=bs)
            assert (ito, rno) == (itr, rnr) and np.array_equal(xo, xr)


def test_golden_and_live_minres(oracle):
    g = gold("bicg.npz")
    rp, ci, v, rhs = g["m_row_ptrs"], g["m_cols"], g["m_vals"], g["m_rhs"]
    for bs, pre in ((0, None), (1, "scalar")):
        x, it, rn = oracle.krylov_solve("minres", rp, ci, v, rhs, max_iters=400, reduction=1e-9, precond=pre)
        assert (it, rn) == tuple(g[f"minres_{bs}_it_rn"]) and np.array_equal(x, g[f"minres_{bs}_x"])
    x, it, rn = oracle.krylov_solve("minres", rp, ci, v, rhs, x0=np.full(len(rhs), 0.5), max_iters=7,
                                    reduction=1e-30, baseline="initial_resnorm")
    assert (it, rn) == tuple(g["minres_lim_it_rn"]) and np.array_equal(x, g["minres_lim_x"])
    if os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libgko_ref_shim.so")):
        ref = _ref()
        rp, ci, v = oracle.stencil_csr(3, 6)
        h = ref.CsrHandle("reference", rp, ci, v)
        b = np.random.default_rng(3).uniform(-1, 1, len(rp) - 1)
        for bs in (0, 1, 8):
            pre = None if bs == 0 else ("scalar" if bs == 1 else "block")
            xo, ito, rno = oracle.krylov_solve("minres", rp, ci, v, b, max_iters=200, reduction=1e-10,
                                               precond=pre, max_block_size=max(bs, 1))
            xr, itr, rnr = h.krylov_solve("minres", b, max_iters=200, reduction=1e-10, precond_block_size=bs)
            assert (ito, rno) == (itr, rnr) and np.array_equal(xo, xr)


# ------------------------------------------------- device_matrix_data assembly
def test_golden_assembly(oracle):
    g = gold("assembly.npz")
    rows, cols, vals = g["rows"], g["cols"], g["vals"]
    s = oracle.md_sort_row_major(rows, cols, vals)
    for got, name in zip(s, ("rows", "cols", "vals")):
        assert got.tobytes() == g["sort_row_major_" + name].tobytes()
    for got, name in zip(oracle.md_remove_zeros(rows, cols, vals), ("rows", "cols", "vals")):
        assert got.tobytes() == g["remove_zeros_" + name].tobytes()
    for got, name in zip(oracle.md_sum_duplicates(*s), ("rows", "cols", "vals")):
        assert got.tobytes() == g["sum_duplicates_" + name].tobytes()
    # Csr::read(device_matrix_data) = the sorted arrays + convert_idxs_to_ptrs
    n_rows = int(g["shape"][0])
    assert np.array_equal(g["csr_rows"], np.concatenate(([0], np.cumsum(np.bincount(s[0], minlength=n_rows)))))
    assert g["csr_cols"].tobytes() == s[1].tobytes() and g["csr_vals"].tobytes() == s[2].tobytes()


def test_assembly_known_properties(oracle):
    # stable: equal (row, column) keep their input order (std::stable_sort,
    # reference/base/device_matrix_data_kernels.cpp:135-143)
    rows = np.array([1, 0, 1, 0, 1], np.int32)
    cols = np.array([2, 5, 2, 5, 0], np.int32)
    vals = np.array([1., 2., 3., 4., 5.])
    r, c, v = oracle.md_sort_row_major(rows, cols, vals)
    assert (r.tolist(), c.tolist(), v.tolist()) == ([0, 0, 1, 1, 1], [5, 5, 0, 2, 2], [2., 4., 5., 1., 3.])
    r, c, v = oracle.md_sum_duplicates(r, c, v)
    assert (r.tolist(), c.tolist(), v.tolist()) == ([0, 1, 1], [5, 0, 2], [6., 5., 4.])
    # a run is summed from 0: -0 alone becomes +0 once ANY run is merged, stays -0 otherwise
    r, c, v = oracle.md_sum_duplicates(np.array([0, 1, 1], np.int32), np.array([0, 1, 1], np.int32),
                                       np.array([-0.0, 1.0, 2.0]))
    assert not np.signbit(v[0]) and v[1] == 3.0
    _, _, v = oracle.md_sum_duplicates(np.array([0, 1], np.int32), np.array([0, 1], np.int32), np.array([-0.0, 1.0]))
    assert np.signbit(v[0])
    # remove_zeros: -0 == 0 goes, NaN stays
    _, _, v = oracle.md_remove_zeros(np.arange(4, dtype=np.int32), np.arange(4, dtype=np.int32),
                                     np.array([0.0, -0.0, np.nan, 3.0]))
    assert len(v) == 2 and np.isnan(v[0]) and v[1] == 3.0


@pytest.mark.parametrize("seed", [1, 2])
def test_live_reference_assembly(oracle, seed):
    ref = _ref()
    rng = np.random.default_rng(seed)
    nnz = 3000
    rows = rng.integers(0, 100, nnz).astype(np.int32)
    cols = rng.integers(0, 30 * seed, nnz).astype(np.int32)
    vals = rng.uniform(1, 2, nnz)
    vals[rng.random(nnz) < 0.3] = 0.0
    s = oracle.md_sort_row_major(rows, cols, vals)
    for a, b in zip(s, ref.md_assemble("sort_row_major", 100, 200, rows, cols, vals)):
        assert a.tobytes() == b.tobytes()
    for a, b in zip(oracle.md_remove_zeros(rows, cols, vals), ref.md_assemble("remove_zeros", 100, 200, rows, cols, vals)):
        assert a.tobytes() == b.tobytes()
    for a, b in zip(oracle.md_sum_duplicates(*s), ref.md_assemble("sum_duplicates", 100, 200, rows, cols, vals)):
        assert a.tobytes() == b.tobytes()
