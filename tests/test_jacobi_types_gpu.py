"""Block-Jacobi with a fixed reduced, autodetected or block-wise storage precision for the value types
float, complex<float>, complex<double> (gkoc_jacobi_{generate,apply,transpose}_adaptive_{f32,c64,c128}_i32)
THROUGH THE C ABI against tests/golden/jacobi_types.npz - what the unmodified reference computes on
gko::ReferenceExecutor (tests/golden/make_jacobi_types_golden.py: reference/preconditioner/
jacobi_kernels.cpp:313-411 generate, :419-531 apply, :528-627 transposes; the rules per component type
core/preconditioner/jacobi_utils.hpp:104-176).
Bit-exact: the blocks found and EVERY block's precision.  Tolerances (relative, Frobenius): condition
numbers 1e-4 / 1e-11, products 1e-5 (float), 1e-4 (complex<float>: blocks kept in half), 1e-12
(complex<double>) - the complex quotient of the inversion is not libstdc++'s bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jacobi_types.npz"))
CASES = sorted({tuple(k.split("/")[:3]) for k in GOLD.files if k.count("/") == 3})
DT = {"f32": (torch.float32, torch.float32, 1e-5, 1e-4), "c64": (torch.complex64, torch.float32, 1e-4, 1e-4),
      "c128": (torch.complex128, torch.float64, 1e-12, 1e-11)}


def _rel(got, want):
    den = np.linalg.norm(want.ravel())
    return np.linalg.norm((got - want).ravel()) / (den if den > 0 else 1.0)


@pytest.mark.parametrize("vt,max_bs,tag", CASES)
def test_precisions_and_products_equal_the_reference(gexec, vt, max_bs, tag):
    from ginkgo_amd._lib import call
    from ginkgo_amd.preconditioner import compute_storage_scheme
    ex = gexec
    dt, rdt, tol, ctol = DT[vt]
    gold = lambda k: GOLD[f"{vt}/{max_bs}/{tag}/{k}"]      # noqa: E731
    rp, ci, vals, b = (GOLD[f"{vt}/{k}"] for k in ("row_ptrs", "col_idxs", "values", "b"))
    n, nrhs, bs = len(rp) - 1, b.shape[1], int(max_bs)
    d_rp, d_ci, d_v, d_b = (ex.to_device(a) for a in (rp, ci, vals, b))
    # the blocks (indices: exact)
    d_bp = ex.zeros((n + 1,), torch.int32)
    nb_c = C.c_int64(0)
    call(f"gkoc_jacobi_find_blocks_{vt}_i32", ex.stream, n, d_rp, d_ci, C.c_uint32(bs), C.byref(nb_c), d_bp)
    nb = nb_c.value
    assert nb == len(gold("block_ptrs")) - 1
    assert np.array_equal(d_bp[:nb + 1].cpu().numpy(), gold("block_ptrs"))
    d_bp = d_bp[:nb + 1].contiguous()
    scheme = compute_storage_scheme(bs, 64)
    gs = 1 << scheme.group_power
    storage = ((nb + gs - 1) // gs) * scheme.group_offset
    blocks = ex.zeros((storage,), dt)
    # the requests replicated over the blocks; in place they become the decisions
    prec = ex.to_device(np.resize(gold("request"), nb).astype(np.uint8))
    cond = ex.zeros((nb,), rdt)
    acc = float(gold("accuracy")[0])
    call(f"gkoc_jacobi_generate_adaptive_{vt}_i32", ex.stream, n, d_rp, d_ci, d_v, nb, C.c_uint32(bs), scheme,
         d_bp, C.c_float(acc) if rdt == torch.float32 else C.c_double(acc), prec, cond, blocks)
    ex.synchronize()
    assert np.array_equal(prec.cpu().numpy(), gold("prec")), (prec.cpu().numpy(), gold("prec"))
    assert _rel(cond.cpu().numpy().astype(np.float64), gold("cond")) < ctol

    def product(blk):
        x = ex.zeros((n, nrhs), dt)
        call(f"gkoc_jacobi_apply_adaptive_{vt}_i32", ex.stream, nb, C.c_uint32(bs), scheme, d_bp, blk, prec, None,
             d_b, nrhs, None, x, nrhs, nrhs)
        ex.synchronize()
        return x.cpu().numpy()

    want = gold("x")
    assert _rel(product(blocks), want) < tol
    # x = alpha M b + beta x
    alpha = ex.to_device(np.asarray([1.5], want.dtype))
    beta = ex.to_device(np.asarray([-0.75], want.dtype))
    x0 = np.ascontiguousarray(b[::-1]).copy()
    x = ex.to_device(x0.copy())
    call(f"gkoc_jacobi_apply_adaptive_{vt}_i32", ex.stream, nb, C.c_uint32(bs), scheme, d_bp, blocks, prec, alpha,
         d_b, nrhs, beta, x, nrhs, nrhs)
    ex.synchronize()
    assert _rel(x.cpu().numpy(), 1.5 * want - 0.75 * x0) < 4 * tol
    # transpose_jacobi / conj_transpose_jacobi: the blocks move in their storage type
    for conj, key in ((0, "xt"), (1, "xh")):
        out = ex.zeros((storage,), dt)
        call(f"gkoc_jacobi_transpose_adaptive_{vt}_i32", ex.stream, nb, scheme, d_bp, blocks, prec, C.c_int(conj),
             out)
        assert _rel(product(out), gold(key)) < tol


@pytest.mark.parametrize("max_bs,tag", [(c[1], c[2]) for c in CASES if c[0] == "f32"])
def test_python_mirror_on_float_values(gexec, max_bs, tag):
    """the same fixture through the host-side mirror of the reference interface:
    Jacobi.build().with_storage_optimization(...).with_accuracy(...).on(exec).generate(A) on a float matrix"""
    import ginkgo_amd as g
    gold = lambda k: GOLD[f"f32/{max_bs}/{tag}/{k}"]      # noqa: E731
    rp, ci, vals, b = (GOLD[f"f32/{k}"] for k in ("row_ptrs", "col_idxs", "values", "b"))
    n, nrhs = len(rp) - 1, b.shape[1]
    a = g.Csr.from_arrays(gexec, (n, n), rp, ci, vals)
    name = lambda r: "autodetect" if r == 0xff else (int(r) >> 4, int(r) & 15)      # noqa: E731
    req = gold("request")
    f = g.Jacobi.build().with_max_block_size(int(max_bs)).with_accuracy(float(gold("accuracy")[0]))
    if len(req) > 1:
        f.with_storage_optimization([name(r) for r in req])
    elif req[0] == 0xff:
        f.with_storage_optimization("autodetect")
    else:
        f.with_storage_optimization(*name(req[0]))
    jac = f.on(gexec).generate(a)
    assert jac.get_num_blocks() == len(gold("prec"))
    assert np.array_equal(jac.block_pointers.cpu().numpy(), gold("block_ptrs"))
    assert np.array_equal(jac.precisions.cpu().numpy(), gold("prec"))
    assert _rel(jac.conditioning.cpu().numpy().astype(np.float64), gold("cond")) < 1e-4
    db = g.Dense.from_numpy(gexec, b)
    for op, key in ((jac, "x"), (jac.transpose(), "xt"), (jac.conj_transpose(), "xh")):
        x = g.Dense.from_numpy(gexec, np.zeros((n, nrhs), np.float32))
        op.apply(db, x)
        assert x.to_numpy().dtype == np.float32 and _rel(x.to_numpy(), gold(key)) < 1e-5, key


def _problem(vt, max_bs, n_blocks, seed):
    """diagonal blocks of 1 .. max_bs rows, from well conditioned to nearly singular (so that the
    autodetection lands on several storage types), a few entries outside the blocks"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    np_dt = {"f32": np.float32, "c64": np.complex64, "c128": np.complex128}[vt]
    sizes = rng.integers(1, max_bs + 1, n_blocks)
    n = int(sizes.sum())
    a = sp.lil_matrix((n, n), dtype=np.complex128)
    s = 0
    for k, bs in enumerate(sizes):
        blk = rng.uniform(-1, 1, (bs, bs)) + 1j * rng.uniform(-1, 1, (bs, bs))
        blk += np.eye(bs) * (bs, 1.0, 0.05)[k % 3]
        a[s:s + bs, s:s + bs] = blk
        if s + bs < n:
            a[s, n - 1 - (s % 7)] = 0.01
        s += bs
    a = a.tocsr()
    a.sort_indices()
    vals = a.data if vt != "f32" else a.data.real
    b = rng.uniform(-1, 1, (n, 2)) + (1j * rng.uniform(-1, 1, (n, 2)) if vt != "f32" else 0)
    return a.indptr.astype(np.int32), a.indices.astype(np.int32), np.ascontiguousarray(vals.astype(np_dt)), \
        np.ascontiguousarray(b.astype(np_dt))


@pytest.mark.parametrize("max_bs", [5, 16, 29])
@pytest.mark.parametrize("vt", ["f32", "c64", "c128"])
def test_larger_problems_against_the_oracle(gexec, oracle, vt, max_bs):
    """~900 blocks, autodetect and a block-wise mix, HIP through the C ABI against the oracle restatement
    (oracle/gko_oracle_jacobi_types.inc, itself pinned bit for bit by the fixture): blocks and precisions exact
    for all three value types; float is BIT-IDENTICAL throughout (same operations in the same order, like the
    double path); complex values agree to rounding (the quotient, csrc/complex_type.hpp)"""
    from ginkgo_amd._lib import call
    from ginkgo_amd.preconditioner import compute_storage_scheme
    ex = gexec
    dt, rdt, tol, ctol = DT[vt]
    rp, ci, vals, b = _problem(vt, max_bs, 900, 1000 * max_bs + len(vt))
    n, nrhs = len(rp) - 1, b.shape[1]
    d_rp, d_ci, d_v, d_b = (ex.to_device(a) for a in (rp, ci, vals, b))
    d_bp = ex.zeros((n + 1,), torch.int32)
    nb_c = C.c_int64(0)
    call(f"gkoc_jacobi_find_blocks_{vt}_i32", ex.stream, n, d_rp, d_ci, C.c_uint32(max_bs), C.byref(nb_c), d_bp)
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, max_bs)
    assert nb_c.value == nb and np.array_equal(d_bp[:nb + 1].cpu().numpy(), ptrs[:nb + 1])
    d_bp = d_bp[:nb + 1].contiguous()
    scheme = compute_storage_scheme(max_bs, 64)
    oscheme = oracle.jacobi_storage_scheme(max_bs)
    assert (scheme.block_offset, scheme.group_offset, scheme.group_power) == tuple(oscheme)
    gs = 1 << scheme.group_power
    storage = ((nb + gs - 1) // gs) * scheme.group_offset
    bitwise = []
    for request, acc in (([0xff], 1e-1), ([0xff], 1e-3), ([0xff, 0x01, 0xff, 0x20, 0xff, 0x00, 0x11, 0xff], 1e-2)):
        want_blocks, want_prec, want_cond = oracle.jacobi_generate_adaptive_t(rp, ci, vals, nb, oscheme, ptrs[:nb + 1],
                                                                              acc, request)
        blocks = ex.zeros((storage,), dt)
        prec = ex.to_device(np.resize(np.asarray(request, np.uint8), nb))
        cond = ex.zeros((nb,), rdt)
        call(f"gkoc_jacobi_generate_adaptive_{vt}_i32", ex.stream, n, d_rp, d_ci, d_v, nb, C.c_uint32(max_bs), scheme,
             d_bp, C.c_float(acc) if rdt == torch.float32 else C.c_double(acc), prec, cond, blocks)
        ex.synchronize()
        got_prec = prec.cpu().numpy()
        differ = got_prec != want_prec
        # float: the same decisions, block by block.  Complex: the condition numbers agree to rounding times
        # the condition number, so a block within that distance of a threshold may be decided the other way
        # (none in these problems so far; a handful would not be an error, more than 1 % would)
        assert not differ.any() if vt == "f32" else differ.mean() <= 0.01, np.nonzero(differ)[0][:10]
        same_rows = np.repeat(~differ, np.diff(ptrs[:nb + 1]))
        if len(request) == 1 and acc == 1e-1:
            assert len(set(want_prec.tolist())) >= 2      # the problem exercises more than one storage type
        got_cond = cond.cpu().numpy()
        assert _rel(got_cond.astype(np.float64), want_cond.astype(np.float64)) < 50 * ctol
        x = ex.zeros((n, nrhs), dt)
        call(f"gkoc_jacobi_apply_adaptive_{vt}_i32", ex.stream, nb, C.c_uint32(max_bs), scheme, d_bp, blocks, prec,
             None, d_b, nrhs, None, x, nrhs, nrhs)
        ex.synchronize()
        want_x = oracle.jacobi_apply_adaptive_t(nb, oscheme, ptrs[:nb + 1], want_blocks, want_prec, b)
        assert _rel(x.cpu().numpy()[same_rows], want_x[same_rows]) < 20 * tol      # (nearly singular blocks)
        bitwise.append((got_cond.tobytes() == want_cond.tobytes(), x.cpu().numpy().tobytes() == want_x.tobytes()))
    if vt == "f32":
        assert all(c and x_ for c, x_ in bitwise), bitwise


@pytest.mark.parametrize("max_bs", [1, 3, 8, 13, 32])
@pytest.mark.parametrize("vt", ["f32", "c64", "c128"])
def test_lane_layout_equals_the_thread_per_row_kernels_bit_for_bit(gexec, vt, max_bs):
    """round 6: jacobi_apply_lanes_any_kernel (lane = (block, row) of a storage group, blocks in registers,
    b by shuffle) against the round-5 thread-per-row kernels (GKOC_TUNE_JACOBI_LANES = 1) on the same blocks:
    adaptive storage (x = M b, x = alpha M b + beta x, beta = 0 with NaN in x) and, for complex values, the
    full-storage simple_apply / apply - the same operations in the same order, so the same bits."""
    from ginkgo_amd._lib import call, lib
    from ginkgo_amd.preconditioner import compute_storage_scheme
    ex = gexec
    dt, rdt, _, _ = DT[vt]
    rp, ci, vals, b = _problem(vt, max_bs, 1500, 77 * max_bs + len(vt))
    n, nrhs = len(rp) - 1, b.shape[1]
    d_rp, d_ci, d_v, d_b = (ex.to_device(a) for a in (rp, ci, vals, b))
    d_bp = ex.zeros((n + 1,), torch.int32)
    nb_c = C.c_int64(0)
    call(f"gkoc_jacobi_find_blocks_{vt}_i32", ex.stream, n, d_rp, d_ci, C.c_uint32(max_bs), C.byref(nb_c), d_bp)
    nb = nb_c.value
    d_bp = d_bp[:nb + 1].contiguous()
    scheme = compute_storage_scheme(max_bs, 64)
    gs = 1 << scheme.group_power
    storage = ((nb + gs - 1) // gs) * scheme.group_offset
    blocks = ex.zeros((storage,), dt)
    prec = ex.to_device(np.resize(np.asarray([0xff, 0x01, 0xff, 0x20, 0xff, 0x00, 0x11, 0x02], np.uint8), nb))
    cond = ex.zeros((nb,), rdt)
    call(f"gkoc_jacobi_generate_adaptive_{vt}_i32", ex.stream, n, d_rp, d_ci, d_v, nb, C.c_uint32(max_bs), scheme,
         d_bp, C.c_float(1e-2) if rdt == torch.float32 else C.c_double(1e-2), prec, cond, blocks)
    full = ex.zeros((storage,), dt)
    if vt != "f32":
        call(f"gkoc_jacobi_generate_{vt}_i32", ex.stream, n, d_rp, d_ci, d_v, nb, C.c_uint32(max_bs), scheme, d_bp,
             full, None)
    np_dt = b.dtype
    alpha = ex.to_device(np.asarray([1.5], np_dt))
    beta = ex.to_device(np.asarray([-0.75], np_dt))
    zero = ex.to_device(np.asarray([0.0], np_dt))
    x0 = np.ascontiguousarray(b[::-1]).copy()

    def products():
        out = []
        for al, be, start in ((None, None, None), (alpha, beta, x0), (alpha, zero, np.full_like(x0, np.nan))):
            x = ex.zeros((n, nrhs), dt) if start is None else ex.to_device(start.copy())
            call(f"gkoc_jacobi_apply_adaptive_{vt}_i32", ex.stream, nb, C.c_uint32(max_bs), scheme, d_bp, blocks,
                 prec, al, d_b, nrhs, be, x, nrhs, nrhs)
            out.append(x.cpu().numpy().tobytes())
            # one column with a stride (ldb = nrhs, one right-hand side)
            x1 = ex.zeros((n, nrhs), dt) if start is None else ex.to_device(start.copy())
            call(f"gkoc_jacobi_apply_adaptive_{vt}_i32", ex.stream, nb, C.c_uint32(max_bs), scheme, d_bp, blocks,
                 prec, al, d_b, nrhs, be, x1, nrhs, 1)
            out.append(x1.cpu().numpy().tobytes())
            if vt != "f32":
                x2 = ex.zeros((n, nrhs), dt) if start is None else ex.to_device(start.copy())
                if al is None:
                    call(f"gkoc_jacobi_simple_apply_{vt}_i32", ex.stream, nb, C.c_uint32(max_bs), scheme, d_bp, full,
                         d_b, nrhs, x2, nrhs, nrhs)
                else:
                    call(f"gkoc_jacobi_apply_{vt}_i32", ex.stream, nb, C.c_uint32(max_bs), scheme, d_bp, full, al,
                         d_b, nrhs, be, x2, nrhs, nrhs)
                out.append(x2.cpu().numpy().tobytes())
        ex.synchronize()
        return out

    key = 15    # GKOC_TUNE_JACOBI_LANES
    try:
        lib().gkoc_tune_set(C.c_int(key), C.c_int64(1))
        old = products()
        lib().gkoc_tune_set(C.c_int(key), C.c_int64(0))
        new = products()
    finally:
        lib().gkoc_tune_set(C.c_int(key), C.c_int64(0))
    assert [a == b_ for a, b_ in zip(old, new)] == [True] * len(old)
    assert not np.isnan(np.frombuffer(new[4], dtype=np_dt).view(rdt_np(rdt))).any()


def rdt_np(rdt):
    return np.float32 if rdt == torch.float32 else np.float64
