"""Complex value types (include/ginkgo/core/base/types.hpp:471, 689) through the C ABI: the Krylov
step kernels, the GMRES kernels and the ELL / SELL-P products for complex<float> / complex<double>
are the real templates instantiated on gkoc_cplx (csrc/complex_type.hpp).  The reference here is
numpy's complex arithmetic on the same expressions (reference/solver/{cg,bicgstab,fcg,pipe_cg,
gmres,common_gmres}_kernels.cpp, reference/matrix/{ell,sellp}_kernels.cpp) - there is no complex
restatement in oracle/.  Tolerance: r<value_type> of the reference's own tests (1e-14 for
complex<double>, 1e-6 for complex<float>, relative to the largest entry): complex kernels agree to
rounding, not bit for bit (textbook product, Smith quotient)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CT = {"c128": (np.complex128, np.float64, 1e-14), "c64": (np.complex64, np.float32, 2e-6)}


def _dev(gexec, a):
    import torch
    return gexec.to_device(torch.from_numpy(np.ascontiguousarray(a)))


def _host(t):
    return t.cpu().numpy()


def _close(got, want, tol):
    scale = max(1.0, float(np.max(np.abs(want)))) if want.size else 1.0
    assert np.max(np.abs(got - want)) <= tol * scale if want.size else True, np.max(np.abs(got - want))


def _rand(rng, shape, ct):
    return (rng.uniform(-1, 1, shape) + 1j * rng.uniform(-1, 1, shape)).astype(ct)


@pytest.mark.parametrize("tn", ["c128", "c64"])
@pytest.mark.parametrize("rows,cols", [(1000, 1), (777, 3)])
def test_cg_steps_complex(gexec, tn, rows, cols):
    """cg::step_1 / step_2 (reference/solver/cg_kernels.cpp:53-100) incl. a stopped column and a zero
    prev_rho / beta"""
    from ginkgo_amd._lib import call
    import torch
    ct, rt, tol = CT[tn]
    rng = np.random.default_rng(rows + cols)
    p, z, x, r, q = (_rand(rng, (rows, cols), ct) for _ in range(5))
    rho, prev_rho, beta = (_rand(rng, (cols,), ct) for _ in range(3))
    stop = np.zeros(cols, np.uint8)
    if cols > 1:
        stop[1] = 0x41
        prev_rho[2] = 0
        beta[2] = 0
    dp, dz, dx, dr, dq = (_dev(gexec, v) for v in (p, z, x, r, q))
    drho, dprev, dbeta, dstop = (_dev(gexec, v) for v in (rho, prev_rho, beta, stop))
    call("gkoc_cg_step_1_" + tn, gexec.stream, rows, cols, dp, cols, dz, cols, drho, dprev, dstop)
    call("gkoc_cg_step_2_" + tn, gexec.stream, rows, cols, dx, cols, dr, cols, dp, cols, dq, cols, dbeta, drho,
         dstop)
    torch.cuda.synchronize()
    wp, wx, wr = p.copy(), x.copy(), r.copy()
    for j in range(cols):
        if stop[j] & 0x3f:
            continue
        wp[:, j] = z[:, j] if prev_rho[j] == 0 else z[:, j] + (rho[j] / prev_rho[j]) * p[:, j]
    for j in range(cols):
        if stop[j] & 0x3f or beta[j] == 0:
            continue
        t = rho[j] / beta[j]
        wx[:, j] = x[:, j] + t * wp[:, j]
        wr[:, j] = r[:, j] - t * q[:, j]
    _close(_host(dp), wp, tol)
    _close(_host(dx), wx, tol)
    _close(_host(dr), wr, tol)


@pytest.mark.parametrize("tn", ["c128", "c64"])
def test_bicgstab_and_pipe_cg_steps_complex(gexec, tn):
    """bicgstab::step_1 (p = r + rho / prev_rho * alpha / omega * (p - omega v),
    reference/solver/bicgstab_kernels.cpp:62-85) with tiny scalars - the case whose textbook complex
    quotient underflowed in float - and pipe_cg::step_2's beta = delta - |rho / prev_rho|^2 beta
    (reference/solver/pipe_cg_kernels.cpp:120-164)"""
    from ginkgo_amd._lib import call
    import torch
    ct, rt, tol = CT[tn]
    n = 513
    rng = np.random.default_rng(3)
    r, p, v = (_rand(rng, (n, 1), ct) for _ in range(3))
    tiny = rt(1e-20) if tn == "c64" else rt(1e-160)
    rho, prev_rho, alpha, omega = (np.array([s], ct) for s in (tiny * (1 + 2j), tiny * (2 - 1j), 0.5 + 0.25j,
                                                               tiny * (1 - 1j)))
    stop = np.zeros(1, np.uint8)
    dr, dp, dv = (_dev(gexec, a) for a in (r, p, v))
    d = [_dev(gexec, a) for a in (rho, prev_rho, alpha, omega, stop)]
    call("gkoc_bicgstab_step_1_" + tn, gexec.stream, n, 1, dr, 1, dp, 1, dv, 1, d[0], d[1], d[2], d[3], d[4])
    torch.cuda.synchronize()
    hi = np.complex128
    tmp = hi(rho[0]) / hi(prev_rho[0]) * hi(alpha[0]) / hi(omega[0])
    want = (r.astype(hi) + tmp * (p.astype(hi) - hi(omega[0]) * v.astype(hi))).astype(ct)
    got = _host(dp)
    assert np.all(np.isfinite(got.view(rt)))
    _close(got, want, 20 * tol * max(1.0, abs(tmp)))
    # pipe_cg::step_2
    z, w, m, nn, pp, q, f, g = (_rand(rng, (n, 1), ct) for _ in range(8))
    prev_rho, rho, delta, beta = (_rand(rng, (1,), ct) for _ in range(4))
    dv8 = [_dev(gexec, a) for a in (pp, q, f, g, z, w, m, nn)]
    ds = [_dev(gexec, a) for a in (beta, prev_rho, rho, delta, stop)]
    call("gkoc_pipe_cg_step_2_" + tn, gexec.stream, n, 1, ds[0], dv8[0], 1, dv8[1], 1, dv8[2], 1, dv8[3], 1,
         dv8[4], 1, dv8[5], 1, dv8[6], 1, dv8[7], 1, ds[1], ds[2], ds[3], ds[4])
    torch.cuda.synchronize()
    t = rho[0] / prev_rho[0]
    wbeta = delta[0] - abs(t) * abs(t) * beta[0]
    _close(_host(ds[0]), np.array([wbeta], ct), 4 * tol)
    _close(_host(dv8[0]), z + t * pp, 4 * tol)
    _close(_host(dv8[3]), nn + t * g, 4 * tol)


@pytest.mark.parametrize("tn", ["c128", "c64"])
def test_gmres_kernels_complex(gexec, tn):
    """gmres::multi_dot = conj(basis) . next, common_gmres::hessenberg_qr with complex Givens rotations
    (reference/solver/gmres_kernels.cpp:74-92, common_gmres_kernels.cpp:29-113): after the rotation the
    subdiagonal entry is zero, |cos|^2 + |sin|^2 = 1, and the residual norm is |sin| times the old one"""
    from ginkgo_amd._lib import call, lib
    import torch
    ct, rt, tol = CT[tn]
    n, k, dots = 3000, 2, 3
    rng = np.random.default_rng(8)
    basis = _rand(rng, ((dots + 1) * n, k), ct)
    nxt = _rand(rng, (n, k), ct)
    dbasis, dnext = _dev(gexec, basis), _dev(gexec, nxt)
    hcol = gexec.zeros((dots + 1, k), torch.complex128 if tn == "c128" else torch.complex64)
    lib().gkoc_gmres_multi_dot_workspace_bytes.restype = C.c_size_t
    nbytes = lib().gkoc_gmres_multi_dot_workspace_bytes(C.c_int64(n), C.c_int64(k), C.c_int64(dots),
                                                        C.c_size_t(np.dtype(ct).itemsize))
    work = gexec.zeros((nbytes,), torch.uint8)
    call("gkoc_gmres_multi_dot_" + tn, gexec.stream, n, k, dots, dbasis, k, dnext, k, hcol, k, work,
         C.c_size_t(nbytes))
    torch.cuda.synchronize()
    want = np.stack([np.sum(np.conj(basis[d * n:(d + 1) * n]) * nxt, axis=0) for d in range(dots)])
    _close(_host(hcol)[:dots], want.astype(ct), 50 * tol * n ** 0.5)
    # hessenberg_qr at iteration 0: h = [h0, h1]^T per column
    h = _rand(rng, (2, k), ct)
    rnc = np.zeros((2, k), ct)
    rnc[0] = rt(3.0)
    gsin, gcos = np.zeros((4, k), ct), np.zeros((4, k), ct)
    rnorm = np.zeros((k,), rt)
    fin = np.zeros((k,), np.int64)          # size_type array, carried as int64
    stop = np.zeros((k,), np.uint8)
    dh, drnc, dsin, dcos, drn, dfin, dstop = (_dev(gexec, a) for a in (h, rnc, gsin, gcos, rnorm, fin, stop))
    call("gkoc_common_gmres_hessenberg_qr_" + tn, gexec.stream, k, dsin, k, dcos, k, drn, drnc, k, dh, k, 0,
         dfin, dstop)
    torch.cuda.synchronize()
    hh, c, s = _host(dh), _host(dcos)[0], _host(dsin)[0]
    hyp = np.sqrt(np.abs(h[0]) ** 2 + np.abs(h[1]) ** 2)
    _close(c, np.conj(h[0]) / hyp, 8 * tol)
    _close(s, np.conj(h[1]) / hyp, 8 * tol)
    _close(hh[0], hyp.astype(ct), 8 * tol * float(np.max(hyp)))
    assert np.all(hh[1] == 0)
    _close(_host(drn), (np.abs(s) * 3.0).astype(rt), 8 * tol * 3)
    assert list(_host(dfin)) == [1] * k


@pytest.mark.parametrize("tn", ["c128", "c64"])
@pytest.mark.parametrize("nrhs", [1, 3])
def test_ell_and_sellp_products_complex(gexec, tn, nrhs):
    """ell::{spmv, advanced_spmv}, sellp::{...} on complex values against scipy's product of the same
    matrix (padding entries skipped, beta = 0 does not read c: c starts as NaN)"""
    import scipy.sparse as sp
    import torch
    from ginkgo_amd._lib import call
    ct, rt, tol = CT[tn]
    rng = np.random.default_rng(21)
    n, m = 300, 257
    a = sp.random(n, m, density=0.04, random_state=rng, format="csr", dtype=np.float64)
    a = (a + 1j * a.multiply(rng.uniform(-1, 1))).tocsr().astype(ct)
    a.sort_indices()
    lens = np.diff(a.indptr)
    per_row = int(lens.max())
    b = _rand(rng, (m, nrhs), ct)
    want = (a.astype(np.complex128) @ b.astype(np.complex128))
    # ELL (column major, stride n + 5)
    stride = n + 5
    cols = -np.ones((per_row, stride), np.int32)
    vals = np.zeros((per_row, stride), ct)
    for r in range(n):
        k = a.indptr[r + 1] - a.indptr[r]
        cols[:k, r] = a.indices[a.indptr[r]:a.indptr[r + 1]]
        vals[:k, r] = a.data[a.indptr[r]:a.indptr[r + 1]]
    dcols, dvals, db = _dev(gexec, cols), _dev(gexec, vals), _dev(gexec, b)
    c = _dev(gexec, np.full((n, nrhs), np.nan + 0j, ct))
    call("gkoc_ell_spmv_" + tn + "_i32", gexec.stream, n, m, per_row, stride, dcols, dvals, db, nrhs, c, nrhs, nrhs)
    torch.cuda.synchronize()
    _close(_host(c), want.astype(ct), 30 * tol)
    alpha, beta = np.array([0.5 - 2j], ct), np.array([0], ct)
    c.copy_(torch.from_numpy(np.full((n, nrhs), np.nan + 0j, ct)))
    call("gkoc_ell_advanced_spmv_" + tn + "_i32", gexec.stream, n, m, per_row, stride, _dev(gexec, alpha), dcols,
         dvals, db, nrhs, _dev(gexec, beta), c, nrhs, nrhs)
    torch.cuda.synchronize()
    _close(_host(c), (alpha[0] * want).astype(ct), 60 * tol)
    c0 = _rand(rng, (n, nrhs), ct)
    beta = np.array([-1 + 0.5j], ct)
    c.copy_(torch.from_numpy(c0))
    call("gkoc_ell_advanced_spmv_" + tn + "_i32", gexec.stream, n, m, per_row, stride, _dev(gexec, alpha), dcols,
         dvals, db, nrhs, _dev(gexec, beta), c, nrhs, nrhs)
    torch.cuda.synchronize()
    _close(_host(c), (alpha[0] * want + beta[0] * c0).astype(ct), 60 * tol)
    # SELL-P, slice size 64
    ss = 64
    n_slices = -(-n // ss)
    slens = np.array([lens[s * ss:(s + 1) * ss].max() for s in range(n_slices)], np.uint64)
    sets = np.concatenate([[0], np.cumsum(slens)]).astype(np.uint64)
    total = int(sets[-1]) * ss
    scols = -np.ones(total, np.int32)
    svals = np.zeros(total, ct)
    for r in range(n):
        s_, ir = divmod(r, ss)
        for k in range(a.indptr[r + 1] - a.indptr[r]):
            idx = (int(sets[s_]) + k) * ss + ir
            scols[idx] = a.indices[a.indptr[r] + k]
            svals[idx] = a.data[a.indptr[r] + k]
    c.copy_(torch.from_numpy(np.full((n, nrhs), np.nan + 0j, ct)))
    call("gkoc_sellp_spmv_" + tn + "_i32", gexec.stream, n, m, ss, _dev(gexec, sets.view(np.int64)),
         _dev(gexec, slens.view(np.int64)), _dev(gexec, scols), _dev(gexec, svals), db, nrhs, c, nrhs, nrhs)
    torch.cuda.synchronize()
    _close(_host(c), want.astype(ct), 30 * tol)
    c.copy_(torch.from_numpy(c0))
    call("gkoc_sellp_advanced_spmv_" + tn + "_i32", gexec.stream, n, m, ss, _dev(gexec, alpha),
         _dev(gexec, sets.view(np.int64)), _dev(gexec, slens.view(np.int64)), _dev(gexec, scols),
         _dev(gexec, svals), db, nrhs, _dev(gexec, beta), c, nrhs, nrhs)
    torch.cuda.synchronize()
    _close(_host(c), (alpha[0] * want + beta[0] * c0).astype(ct), 60 * tol)


@pytest.mark.parametrize("tn", ["c128", "c64"])
@pytest.mark.parametrize("it", ["i32", "i64"])
@pytest.mark.parametrize("max_bs,nrhs", [(4, 1), (13, 3), (32, 2)])
def test_block_jacobi_complex(gexec, tn, it, max_bs, nrhs):
    """jacobi::find_blocks / generate / simple_apply / apply on complex values
    (reference/preconditioner/jacobi_kernels.cpp:130-190, 340-411, 413-520): natural blocks of a block
    diagonal matrix with some coupling outside the blocks, the inverse of every diagonal block (numpy's
    inv) applied to b, and x = alpha M b + beta x.  Pivoting is by magnitude, so a block whose leading
    entry is zero is part of the case."""
    import scipy.sparse as sp
    import torch
    from ginkgo_amd._lib import call
    from ginkgo_amd.preconditioner import compute_storage_scheme
    ct, rt, tol = CT[tn]
    idt = np.int32 if it == "i32" else np.int64
    rng = np.random.default_rng(7 * max_bs + nrhs)
    sizes = rng.integers(1, max_bs + 1, 41)
    sizes[0] = max_bs
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    n = int(ptr[-1])
    dense_blocks = []
    for k, sz in enumerate(sizes):
        blk = _rand(rng, (sz, sz), ct) + (2.0 * sz) * np.eye(sz, dtype=ct)
        if k == 3 and sz > 1:
            blk[0, 0] = 0          # forces a row exchange
        dense_blocks.append(blk)
    a = sp.block_diag(dense_blocks, format="lil", dtype=ct)
    for _ in range(30):            # entries outside the blocks are not the preconditioner's business
        i, j = rng.integers(0, n, 2)
        if np.searchsorted(ptr, i, side="right") != np.searchsorted(ptr, j, side="right"):
            a[i, j] = ct(0.25 - 0.5j)
    a = sp.csr_matrix(a)
    a.sort_indices()
    drp, dci, dv = _dev(gexec, a.indptr.astype(idt)), _dev(gexec, a.indices.astype(idt)), _dev(gexec, a.data)
    dptr = _dev(gexec, ptr.astype(idt))
    scheme = compute_storage_scheme(max_bs)
    gs = 1 << scheme.group_power
    nb = len(sizes)
    blocks = gexec.zeros((((nb + gs - 1) // gs) * scheme.group_offset,), torch.from_numpy(np.zeros(1, ct)).dtype)
    suf = f"{tn}_{it}"
    call("gkoc_jacobi_generate_" + suf, gexec.stream, n, drp, dci, dv, nb, C.c_uint32(max_bs), scheme, dptr,
         blocks, None)
    b = _rand(rng, (n, nrhs), ct)
    x0 = _rand(rng, (n, nrhs), ct)
    db, dx = _dev(gexec, b), _dev(gexec, x0)
    call("gkoc_jacobi_simple_apply_" + suf, gexec.stream, nb, C.c_uint32(max_bs), scheme, dptr, blocks, db, nrhs,
         dx, nrhs, nrhs)
    torch.cuda.synchronize()
    want = np.concatenate([np.linalg.solve(blk.astype(np.complex128), b[ptr[k]:ptr[k + 1]].astype(np.complex128))
                           for k, blk in enumerate(dense_blocks)])
    # conditioning of these blocks: < 10, so the inverse applied in working precision stays within
    # a few tens of ulps
    _close(_host(dx), want.astype(ct), 40 * tol)
    alpha, beta = _rand(rng, (1,), ct), _rand(rng, (1,), ct)
    dx2 = _dev(gexec, x0)
    call("gkoc_jacobi_apply_" + suf, gexec.stream, nb, C.c_uint32(max_bs), scheme, dptr, blocks,
         _dev(gexec, alpha), db, nrhs, _dev(gexec, beta), dx2, nrhs, nrhs)
    torch.cuda.synchronize()
    _close(_host(dx2), (alpha[0] * want + beta[0] * x0.astype(np.complex128)).astype(ct), 40 * tol)
    # natural blocks of the block diagonal part alone: agglomerated up to max_bs like the real kernels
    # (same find_blocks_impl; the values play no role) - block pointers ascend, end at n, respect max_bs
    only = sp.csr_matrix(sp.block_diag(dense_blocks, dtype=ct))
    only.sort_indices()
    found = gexec.alloc((n + 1,), torch.int32 if it == "i32" else torch.int64)
    cnt = C.c_int64(0)
    call("gkoc_jacobi_find_blocks_" + suf, gexec.stream, n, _dev(gexec, only.indptr.astype(idt)),
         _dev(gexec, only.indices.astype(idt)), C.c_uint32(max_bs), C.byref(cnt), found)
    torch.cuda.synchronize()
    fp = _host(found)[:cnt.value + 1]
    assert fp[0] == 0 and fp[-1] == n and (np.diff(fp) > 0).all() and (np.diff(fp) <= max_bs).all()
    assert np.isin(fp, ptr).all()      # never through the middle of a dense block
    # max_block_size above 32 is refused, loudly
    from ginkgo_amd._lib import NotSupported
    with pytest.raises(NotSupported):
        call("gkoc_jacobi_generate_" + suf, gexec.stream, n, drp, dci, dv, nb, C.c_uint32(33), scheme, dptr,
             blocks, None)


@pytest.mark.parametrize("tn", ["c128", "c64"])
@pytest.mark.parametrize("it", ["i32", "i64"])
@pytest.mark.parametrize("shape", ["stencil", "ragged"])
def test_csr_spmv_complex_row_segment_kernel(gexec, tn, it, shape):
    """round 6: csr::spmv / advanced_spmv on complex values through the row-segment kernel of the real types
    (csrc/csr_spmv.hip csr_spmv_complex) against scipy on the same matrix (reference/matrix/csr_kernels.cpp:53-112:
    the row sum in storage order) and against round 5's thread-per-row kernel (GKOC_TUNE_CCSR_THREAD_PER_ROW = 1):
    empty rows, a row of 5000 entries (the cooperative path), rows that do not fill the last segment, 1 and 3
    right-hand sides with a stride, beta = 0 on NaN."""
    import scipy.sparse as sp
    import torch
    from ginkgo_amd._lib import call, lib
    ct, rt, tol = CT[tn]
    rng = np.random.default_rng(11 + len(shape))
    if shape == "stencil":
        g = 23
        n = g ** 3
        idx = np.arange(n).reshape(g, g, g)
        rows_l, cols_l = [], []
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    src = idx[max(0, -dz):g - max(0, dz), max(0, -dy):g - max(0, dy), max(0, -dx):g - max(0, dx)]
                    dst = idx[max(0, dz):g - max(0, -dz), max(0, dy):g - max(0, -dy), max(0, dx):g - max(0, -dx)]
                    rows_l.append(src.ravel())
                    cols_l.append(dst.ravel())
        r_, c_ = np.concatenate(rows_l), np.concatenate(cols_l)
        a = sp.csr_matrix((np.ones(len(r_)), (r_, c_)), shape=(n, n))
    else:
        n = 4099
        a = sp.random(n, n, density=0.004, format="lil", random_state=3)
        a[5, :] = 0                      # empty rows
        a[4000, :] = 0
        a[77, rng.choice(n, 3000, replace=False)] = 1.0      # (long for this size)
        a = a.tocsr()
        big = sp.random(1, 20000, density=0.25, format="csr", random_state=4)     # a row of ~5000 entries
        a = sp.vstack([sp.hstack([a, sp.csr_matrix((n, 20000 - n))]), big,
                       sp.csr_matrix((20000 - n - 1, 20000))]).tocsr()
        n = 20000
    a.sort_indices()
    a.data = _rand(rng, a.data.shape, ct)
    a = a.astype(ct)
    itype = np.int32 if it == "i32" else np.int64
    d_rp, d_ci, d_v = _dev(gexec, a.indptr.astype(itype)), _dev(gexec, a.indices.astype(itype)), _dev(gexec, a.data)
    for nrhs in (1, 3):
        ld = nrhs + 1
        x = _rand(rng, (n, ld), ct)
        y0 = _rand(rng, (n, ld), ct)
        alpha, beta = _rand(rng, (1,), ct), _rand(rng, (1,), ct)
        d_x = _dev(gexec, x)
        d_alpha, d_beta, d_zero = _dev(gexec, alpha), _dev(gexec, beta), _dev(gexec, np.zeros(1, ct))
        want = a @ x[:, :nrhs]
        scale = np.abs(a).dot(np.abs(x[:, :nrhs])).max()
        got = {}
        for mode in (1, 0):
            lib().gkoc_tune_set(C.c_int(16), C.c_int64(mode))
            try:
                y = _dev(gexec, y0.copy())
                call(f"gkoc_ccsr_spmv_{tn}_{it}", gexec.stream, n, nrhs, d_rp, d_ci, d_v, None, d_x, ld, None, y, ld)
                ya = _dev(gexec, y0.copy())
                call(f"gkoc_ccsr_spmv_{tn}_{it}", gexec.stream, n, nrhs, d_rp, d_ci, d_v, d_alpha, d_x, ld, d_beta, ya,
                     ld)
                yn = _dev(gexec, np.full_like(y0, np.nan))
                call(f"gkoc_ccsr_spmv_{tn}_{it}", gexec.stream, n, nrhs, d_rp, d_ci, d_v, d_alpha, d_x, ld, d_zero, yn,
                     ld)
                torch.cuda.synchronize()
                got[mode] = (_host(y), _host(ya), _host(yn))
            finally:
                lib().gkoc_tune_set(C.c_int(16), C.c_int64(0))
        for mode in (1, 0):
            y, ya, yn = got[mode]
            assert np.max(np.abs(y[:, :nrhs] - want)) <= 40 * tol * scale, (mode, nrhs)
            assert np.array_equal(y[:, nrhs:], y0[:, nrhs:])                       # the padding column is untouched
            assert np.max(np.abs(ya[:, :nrhs] - (alpha[0] * want + beta[0] * y0[:, :nrhs]))) <= 80 * tol * scale
            assert np.max(np.abs(yn[:, :nrhs] - alpha[0] * want)) <= 80 * tol * scale
        # the plain product: the same sums in the same order => the same bits as the thread-per-row kernel,
        # except on the rows longer than GKOC_CSR_LONG_ROW, which the row-segment kernel sums in 64 chunks
        lens = np.diff(a.indptr)
        short = lens <= 4096
        assert np.array_equal(got[0][0][short, :nrhs], got[1][0][short, :nrhs])
