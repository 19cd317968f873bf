"""GPU parity of the conversions, permutations, SpGEMM / SpGEAM and L1-Jacobi helpers of
csrc/conversions.hip, called through the C ABI, against numpy / scipy restatements of the reference
loops (reference/matrix/{dense,csr,coo,ell,sellp,hybrid}_kernels.cpp, cited in conversions.hip).
Index arrays and copied values bit-exact; sums of products (SpGEMM) to 1e-14.

Ginkgo's own test binaries exercise the same entry points through the binding
(tests/test_reftests_gpu.py); this file needs neither the reference build nor the binding."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from util import random_csr

pytestmark = pytest.mark.gpu

TYPES = [(np.float64, np.int32), (np.float64, np.int64), (np.float32, np.int32)]


def _suf(vt, it):
    return {np.float64: "f64", np.float32: "f32"}[vt] + "_" + {np.int32: "i32", np.int64: "i64"}[it]


def _isuf(it):
    return {np.int32: "i32", np.int64: "i64"}[it]


def _dense_case(rng, rows, cols, dens, vt):
    d = rng.uniform(-1, 1, (rows, cols)).astype(vt)
    d[rng.uniform(0, 1, (rows, cols)) > dens] = 0
    if rows > 3:
        d[2, :] = 0                      # an empty row
    return d


def _scan(gexec, call, t, it):
    call("gkoc_prefix_sum_nonnegative_" + _isuf(it), gexec.stream, t, t.numel())


@pytest.mark.parametrize("vt,it", TYPES)
def test_dense_to_sparse_and_back(gexec, vt, it):
    from ginkgo_amd._lib import call
    rng = np.random.default_rng(1)
    for rows, cols, dens in ((37, 53, 0.2), (130, 7, 0.6), (1, 1, 1.0), (5, 9, 0.0)):
        d = _dense_case(rng, rows, cols, dens, vt)
        ref = sp.csr_matrix(d)
        ref.sort_indices()
        dd = gexec.to_device(d)
        s = _suf(vt, it)
        # count + scan + fill: Csr
        ptrs = gexec.to_device(np.zeros(rows + 1, it))
        call("gkoc_dense_count_nonzeros_per_row_" + s[:3], gexec.stream, rows, cols, dd, cols, ptrs,
             np.dtype(it).itemsize)
        assert np.array_equal(ptrs.cpu().numpy()[:rows], np.diff(ref.indptr))
        _scan(gexec, call, ptrs, it)
        assert np.array_equal(ptrs.cpu().numpy(), ref.indptr)
        nnz = int(ref.nnz)
        ci, v = gexec.to_device(np.zeros(nnz, it)), gexec.to_device(np.zeros(nnz, vt))
        call("gkoc_dense_to_csr_" + s, gexec.stream, rows, cols, dd, cols, ptrs, ci, v)
        assert np.array_equal(ci.cpu().numpy(), ref.indices) and np.array_equal(v.cpu().numpy(), ref.data)
        back = gexec.to_device(np.zeros((rows, cols), vt))
        call("gkoc_csr_fill_in_dense_" + s, gexec.stream, rows, ptrs, ci, v, back, cols)
        assert np.array_equal(back.cpu().numpy(), d)
        # Coo (row pointers in int64, as core/matrix/dense.cpp passes them)
        p64 = gexec.to_device(ref.indptr.astype(np.int64))
        ri = gexec.to_device(np.zeros(nnz, it))
        call("gkoc_dense_to_coo_" + s, gexec.stream, rows, cols, dd, cols, p64, ri, ci, v)
        assert np.array_equal(ri.cpu().numpy(), np.repeat(np.arange(rows), np.diff(ref.indptr)))
        back.zero_()
        call("gkoc_coo_fill_in_dense_" + s, gexec.stream, nnz, ri, ci, v, back, cols)
        assert np.array_equal(back.cpu().numpy(), d)
        # max row length, Ell with a padded stride, and back to Csr / Dense
        mx = C.c_uint64(0)
        call("gkoc_dense_max_nnz_per_row_" + s[:3], gexec.stream, rows, cols, dd, cols, C.byref(mx))
        k = int(np.diff(ref.indptr).max()) if rows else 0
        assert mx.value == k
        stride = rows + 3
        ec = gexec.to_device(np.full(stride * max(k, 1), 7, it))
        ev = gexec.to_device(np.full(stride * max(k, 1), 7, vt))
        call("gkoc_dense_to_ell_" + s, gexec.stream, rows, cols, dd, cols, k, k, stride, ec, ev, None, None,
             None, None)
        ecn, evn = ec.cpu().numpy()[:stride * k].reshape(k, stride), ev.cpu().numpy()[:stride * k].reshape(k, stride)
        for r in range(rows):
            n = ref.indptr[r + 1] - ref.indptr[r]
            assert np.array_equal(ecn[:n, r], ref.indices[ref.indptr[r]:ref.indptr[r + 1]])
            assert np.array_equal(evn[:n, r], ref.data[ref.indptr[r]:ref.indptr[r + 1]])
            assert (ecn[n:, r] == -1).all() and (evn[n:, r] == 0).all()
        assert (ecn[:, rows:] == -1).all() and (evn[:, rows:] == 0).all()      # the padding rows of the stride
        cnt = gexec.to_device(np.zeros(rows + 1, it))
        call("gkoc_ell_count_nonzeros_per_row_" + _isuf(it), gexec.stream, rows, k, stride, ec, cnt)
        _scan(gexec, call, cnt, it)
        assert np.array_equal(cnt.cpu().numpy(), ref.indptr)
        ci2, v2 = gexec.to_device(np.zeros(nnz, it)), gexec.to_device(np.zeros(nnz, vt))
        call("gkoc_ell_to_csr_" + s, gexec.stream, rows, k, stride, ec, ev, cnt, ci2, v2)
        assert np.array_equal(ci2.cpu().numpy(), ref.indices) and np.array_equal(v2.cpu().numpy(), ref.data)
        back.zero_()
        call("gkoc_ell_fill_in_dense_" + s, gexec.stream, rows, k, stride, ec, ev, back, cols)
        assert np.array_equal(back.cpu().numpy(), d)
        # Hybrid: two entries per row in Ell, the rest in Coo
        lim = min(2, cols)
        over = np.maximum(np.diff(ref.indptr) - lim, 0)
        cptr = np.concatenate([[0], np.cumsum(over)]).astype(np.int64)
        hc, hv = gexec.to_device(np.zeros(rows * max(lim, 1), it)), gexec.to_device(np.zeros(rows * max(lim, 1), vt))
        nco = int(cptr[-1])
        cr, cc, cv = (gexec.to_device(np.zeros(max(nco, 1), t)) for t in (it, it, vt))
        call("gkoc_dense_to_ell_" + s, gexec.stream, rows, cols, dd, cols, lim, lim, rows, hc, hv,
             gexec.to_device(cptr), cr, cc, cv)
        want_r, want_c, want_v = [], [], []
        for r in range(rows):
            a, e = ref.indptr[r] + lim, ref.indptr[r + 1]
            if e > a:
                want_r += [r] * (e - a)
                want_c += ref.indices[a:e].tolist()
                want_v += ref.data[a:e].tolist()
        assert cr.cpu().numpy()[:nco].tolist() == want_r and cc.cpu().numpy()[:nco].tolist() == want_c
        assert np.array_equal(cv.cpu().numpy()[:nco], np.array(want_v, vt))
        ellp = gexec.to_device(np.concatenate([[0], np.cumsum(np.minimum(np.diff(ref.indptr), lim))]).astype(it))
        coop = gexec.to_device(cptr.astype(it))
        orp = gexec.to_device(np.zeros(rows + 1, it))
        call("gkoc_hybrid_to_csr_" + s, gexec.stream, rows, lim, rows, hc, hv, cc, cv, ellp, coop, orp, ci2, v2)
        assert np.array_equal(orp.cpu().numpy(), ref.indptr)
        assert np.array_equal(ci2.cpu().numpy(), ref.indices) and np.array_equal(v2.cpu().numpy(), ref.data)
        # Sellp, slice size 4, stride factor 2
        ss, sf = 4, 2
        ns = -(-rows // ss)
        sets = gexec.to_device(np.zeros(ns + 1, np.int64))       # size_type: the same bits
        lens = gexec.to_device(np.zeros(max(ns, 1), np.int64))
        call("gkoc_dense_compute_slice_sets_" + s[:3], gexec.stream, rows, cols, dd, cols, ss, sf, sets, lens)
        rl = np.diff(ref.indptr)
        want_len = [int(-(-rl[i * ss:(i + 1) * ss].max() // sf) * sf) for i in range(ns)]
        assert lens.cpu().numpy()[:ns].tolist() == want_len
        assert sets.cpu().numpy().tolist() == np.concatenate([[0], np.cumsum(want_len)]).tolist()
        tot = int(sum(want_len)) * ss
        sc, sv = gexec.to_device(np.full(max(tot, 1), 9, it)), gexec.to_device(np.full(max(tot, 1), 9, vt))
        call("gkoc_dense_to_sellp_" + s, gexec.stream, rows, cols, dd, cols, ss, sets, sc, sv)
        call("gkoc_sellp_count_nonzeros_per_row_" + _isuf(it), gexec.stream, rows, ss, sets, sc, cnt)
        _scan(gexec, call, cnt, it)
        assert np.array_equal(cnt.cpu().numpy(), ref.indptr)
        call("gkoc_sellp_to_csr_" + s, gexec.stream, rows, ss, sets, sc, sv, cnt, ci2, v2)
        assert np.array_equal(ci2.cpu().numpy(), ref.indices) and np.array_equal(v2.cpu().numpy(), ref.data)
        back.zero_()
        call("gkoc_sellp_fill_in_dense_" + s, gexec.stream, rows, ss, sets, sc, sv, back, cols)
        assert np.array_equal(back.cpu().numpy(), d)
        # diagonals
        nd = min(rows, cols)
        for name, args in (("ell", (k, stride, ec, ev)), ("sellp", (ss, sets, sc, sv)), ("coo", None)):
            dg = gexec.to_device(np.zeros(nd, vt))
            if name == "coo":
                call("gkoc_coo_extract_diagonal_" + s, gexec.stream, nnz, ri, ci, v, dg)
            else:
                call(f"gkoc_{name}_extract_diagonal_" + s, gexec.stream, nd, *args, dg)
            assert np.array_equal(dg.cpu().numpy(), np.diag(d)[:nd]), name
        dg = gexec.to_device(np.zeros(nd, vt))
        call("gkoc_dense_extract_diagonal_" + s[:3], gexec.stream, nd, dd, cols, dg)
        assert np.array_equal(dg.cpu().numpy(), np.diag(d)[:nd])
        miss = C.c_int(0)
        call("gkoc_csr_missing_diagonal_" + _isuf(it), gexec.stream, nd, ptrs, ci, C.byref(miss))
        assert bool(miss.value) == bool((np.diag(d)[:nd] == 0).any())


@pytest.mark.parametrize("vt", [np.float64, np.float32])
def test_dense_utilities(gexec, vt):
    from ginkgo_amd._lib import call
    rng = np.random.default_rng(2)
    d = rng.uniform(-1, 1, (45, 19)).astype(vt)
    s = "f64" if vt == np.float64 else "f32"
    dd = gexec.to_device(d)
    t = gexec.to_device(np.zeros((19, 45), vt))
    call("gkoc_dense_transpose_" + s, gexec.stream, 45, 19, dd, 19, t, 45)
    assert np.array_equal(t.cpu().numpy(), d.T)
    import ginkgo_amd as g
    work = gexec.alloc((g._lib.lib().gkoc_reduction_workspace_bytes(C.c_int64(45), C.c_int64(19), C.c_size_t(8)),),
                       torch.uint8)
    res = gexec.to_device(np.zeros(19, vt))
    call("gkoc_dense_compute_norm1_" + s, gexec.stream, 45, 19, dd, 19, res, work, C.c_size_t(work.numel()))
    want = np.abs(d.astype(np.float64)).sum(axis=0)
    assert np.max(np.abs(res.cpu().numpy() - want) / want) < (1e-14 if vt == np.float64 else 1e-6)
    al, be = gexec.to_device(np.array([0.7], vt)), gexec.to_device(np.array([-1.3], vt))
    m = gexec.to_device(d.copy())
    call("gkoc_dense_add_scaled_identity_" + s, gexec.stream, 45, 19, al, be, m, 19)
    want = d * vt(-1.3)
    want[np.arange(19), np.arange(19)] += vt(0.7)
    assert np.array_equal(m.cpu().numpy(), want)
    diag = rng.uniform(-1, 1, 19).astype(vt)
    for sub in (0, 1):
        m = gexec.to_device(d.copy())
        call("gkoc_dense_add_scaled_diag_" + s, gexec.stream, 19, al, gexec.to_device(diag), m, 19, sub)
        want = d.copy()
        p = vt(0.7) * diag
        want[np.arange(19), np.arange(19)] = want[np.arange(19), np.arange(19)] - p if sub else \
            want[np.arange(19), np.arange(19)] + p
        assert np.array_equal(m.cpu().numpy(), want)
    seq = gexec.to_device(np.zeros(1000, vt))
    call("gkoc_fill_seq_array_" + s, gexec.stream, seq, 1000)
    assert np.array_equal(seq.cpu().numpy(), np.arange(1000, dtype=vt))


@pytest.mark.parametrize("vt,it", TYPES)
def test_permutations(gexec, vt, it):
    from ginkgo_amd._lib import call
    rng = np.random.default_rng(3)
    s = _suf(vt, it)
    rows, cols = 41, 41
    d = rng.uniform(-1, 1, (rows, cols)).astype(vt)
    dd = gexec.to_device(d)
    rp_, cp_ = rng.permutation(rows).astype(it), rng.permutation(cols).astype(it)
    rs_, cs_ = rng.uniform(1, 2, rows).astype(vt), rng.uniform(1, 2, cols).astype(vt)
    rp, cp, rs, cs = (gexec.to_device(a) for a in (rp_, cp_, rs_, cs_))
    out = gexec.to_device(np.zeros((rows, cols), vt))

    def run(a, b, c, e, inv):
        out.zero_()
        call("gkoc_dense_permute_" + s, gexec.stream, rows, cols, dd, cols, out, cols, a, b, c, e, inv)
        return out.cpu().numpy()
    assert np.array_equal(run(rp, cp, None, None, 0), d[rp_][:, cp_])
    w = np.zeros_like(d)
    w[np.ix_(rp_, cp_)] = d
    assert np.array_equal(run(rp, cp, None, None, 1), w)
    assert np.array_equal(run(None, cp, None, None, 0), d[:, cp_])
    w = np.zeros_like(d)
    w[rp_] = d
    assert np.array_equal(run(rp, None, None, None, 1), w)
    assert np.array_equal(run(rp, cp, rs, cs, 0), (rs_[rp_][:, None] * cs_[cp_][None, :]) * d[rp_][:, cp_])
    w = np.zeros_like(d)
    w[np.ix_(rp_, cp_)] = d / (rs_[rp_][:, None] * cs_[cp_][None, :])
    assert np.array_equal(run(rp, cp, rs, cs, 1), w)
    assert np.array_equal(run(rp, None, rs, None, 0), rs_[rp_][:, None] * d[rp_])
    w = np.zeros_like(d)
    w[:, cp_] = d / cs_[cp_][None, :]
    assert np.array_equal(run(None, cp, None, cs, 1), w)
    # permutation helpers
    inv = gexec.to_device(np.zeros(rows, it))
    call("gkoc_permutation_invert_" + _isuf(it), gexec.stream, rows, rp, inv)
    assert np.array_equal(inv.cpu().numpy(), np.argsort(rp_))
    call("gkoc_permutation_compose_" + _isuf(it), gexec.stream, rows, rp, cp, inv)
    assert np.array_equal(inv.cpu().numpy(), rp_[cp_])
    osc = gexec.to_device(np.zeros(rows, vt))
    call("gkoc_scaled_permutation_invert_" + s, gexec.stream, rows, rs, rp, osc, inv)
    assert np.array_equal(inv.cpu().numpy(), np.argsort(rp_))
    assert np.array_equal(osc.cpu().numpy(), vt(1) / rs_[rp_])
    call("gkoc_scaled_permutation_compose_" + s, gexec.stream, rows, rs, rp, cs, cp, osc, inv)
    comb = rp_[cp_]
    want = np.zeros(rows, vt)
    want[comb] = rs_[comb] * cs_[cp_]
    assert np.array_equal(inv.cpu().numpy(), comb) and np.array_equal(osc.cpu().numpy(), want)
    # advanced_row_gather
    gat = rng.uniform(-1, 1, (7, cols)).astype(vt)
    idx = rng.integers(0, rows, 7).astype(it)
    o = gexec.to_device(gat.copy())
    al, be = gexec.to_device(np.array([0.7], vt)), gexec.to_device(np.array([-1.3], vt))
    call("gkoc_dense_advanced_row_gather_" + s, gexec.stream, 7, cols, al, gexec.to_device(idx), dd, cols, be, o,
         cols)
    assert np.array_equal(o.cpu().numpy(), vt(0.7) * d[idx] + vt(-1.3) * gat)
    # Csr
    rpt, ci, v = random_csr(rows, cols, 0.15, 4, it, dtype=vt, empty_rows=(5,))
    a = sp.csr_matrix((v, ci, rpt), shape=(rows, cols))
    drp, dci, dv = (gexec.to_device(x) for x in (rpt, ci, v))
    orp, oci, ov = gexec.to_device(np.zeros(rows + 1, it)), gexec.to_device(np.zeros(len(ci), it)), \
        gexec.to_device(np.zeros(len(ci), vt))

    def crun(a_, inv_, b_, c_, e_, mode):
        call("gkoc_csr_permute_" + s, gexec.stream, rows, drp, dci, dv, a_, inv_, b_, c_, e_, mode, orp, oci, ov)
        m = sp.csr_matrix((ov.cpu().numpy(), oci.cpu().numpy(), orp.cpu().numpy()), shape=(rows, cols))
        return m.toarray()
    da = a.toarray()
    assert np.array_equal(crun(rp, 0, None, None, None, 0), da[rp_])
    w = np.zeros_like(da)
    w[rp_] = da
    assert np.array_equal(crun(rp, 1, None, None, None, 0), w)
    w = np.zeros_like(da)
    w[:, cp_] = da
    assert np.array_equal(crun(None, 0, cp, None, None, 0), w)
    w = np.zeros_like(da)
    w[np.ix_(rp_, cp_)] = da
    assert np.array_equal(crun(rp, 1, cp, None, None, 0), w)
    assert np.array_equal(crun(rp, 0, None, rs, None, 1), (da * rs_[:, None])[rp_])
    w = np.zeros_like(da)
    w[np.ix_(rp_, cp_)] = da
    w = np.where(w != 0, w / (rs_[:, None] * cs_[None, :]), 0).astype(vt)
    assert np.array_equal(crun(rp, 1, cp, rs, cs, 2), w)
    # entry order inside a row is the source row's (the caller sorts afterwards)
    call("gkoc_csr_permute_" + s, gexec.stream, rows, drp, dci, dv, rp, 0, None, None, None, 0, orp, oci, ov)
    r0 = int(rp_[0])
    assert np.array_equal(oci.cpu().numpy()[:rpt[r0 + 1] - rpt[r0]], ci[rpt[r0]:rpt[r0 + 1]])
    # submatrix rows [3, 30), columns [5, 33)
    cnt = gexec.to_device(np.zeros(28, it))
    call("gkoc_csr_count_in_span_" + s, gexec.stream, 27, 3, 5, 33, drp, dci, cnt)
    sub = a[3:30, 5:33].tocsr()
    sub.sort_indices()
    assert np.array_equal(cnt.cpu().numpy()[:27], np.diff(sub.indptr))
    _scan(gexec, call, cnt, it)
    sci, sv = gexec.to_device(np.zeros(sub.nnz, it)), gexec.to_device(np.zeros(sub.nnz, vt))
    call("gkoc_csr_submatrix_" + s, gexec.stream, 27, 3, 5, 33, drp, dci, dv, cnt, sci, sv)
    assert np.array_equal(sci.cpu().numpy(), sub.indices) and np.array_equal(sv.cpu().numpy(), sub.data)
    # add_scaled_identity on a matrix with a full diagonal
    b = (a + sp.identity(rows, dtype=vt, format="csr") * vt(3)).tocsr()
    b.sort_indices()
    bv = gexec.to_device(b.data.astype(vt))
    call("gkoc_csr_add_scaled_identity_" + s, gexec.stream, rows, gexec.to_device(b.indptr.astype(it)),
         gexec.to_device(b.indices.astype(it)), bv, al, be)
    want = b.data.astype(vt) * vt(-1.3)
    isdiag = b.indices == np.repeat(np.arange(rows), np.diff(b.indptr))
    want[isdiag] += vt(0.7)
    assert np.array_equal(bv.cpu().numpy(), want)


@pytest.mark.parametrize("vt,it", TYPES)
def test_spgemm_spgeam_through_triplets(gexec, vt, it):
    """count + expand, then the library's sort_row_major / sum_duplicates / idxs -> ptrs: pattern =
    the reference's (union, ascending columns), values = sums in the order the reference meets them"""
    import ginkgo_amd as g
    from ginkgo_amd._lib import call
    s = _suf(vt, it)
    tol = 1e-14 if vt == np.float64 else 1e-6
    lib = g._lib.lib()

    def to_csr(n_rows, n_cols, total, tr, tc, tv):
        work = gexec.alloc((max(lib.gkoc_sort_row_major_workspace_bytes(C.c_int64(total), C.c_size_t(np.dtype(vt).itemsize),
                                                                    C.c_size_t(np.dtype(it).itemsize)), 1),), torch.uint8)
        call("gkoc_sort_row_major_" + s, gexec.stream, total, tr, tc, tv, work, C.c_size_t(work.numel()))
        w2 = gexec.alloc((max(lib.gkoc_compact_workspace_bytes(C.c_int64(total)), 1),), torch.uint8)
        kept = C.c_int64(0)
        call("gkoc_sum_duplicates_count_" + _isuf(it), gexec.stream, total, tr, tc, w2, C.c_size_t(w2.numel()),
             C.byref(kept))
        k = kept.value
        orow, ocol, oval = (gexec.to_device(np.zeros(max(k, 1), t)) for t in (it, it, vt))
        call("gkoc_sum_duplicates_fill_" + s, gexec.stream, total, tr, tc, tv, w2, orow, ocol, oval)
        ptrs = gexec.to_device(np.zeros(n_rows + 1, it))
        call("gkoc_convert_idxs_to_ptrs_" + _isuf(it), gexec.stream, k, orow, n_rows, ptrs)
        return sp.csr_matrix((oval.cpu().numpy()[:k], ocol.cpu().numpy()[:k], ptrs.cpu().numpy()),
                             shape=(n_rows, n_cols))

    def product(a, b, alpha, beta, d):
        da = [gexec.to_device(x) for x in (a.indptr.astype(it), a.indices.astype(it), a.data.astype(vt))]
        db = [gexec.to_device(x) for x in (b.indptr.astype(it), b.indices.astype(it), b.data.astype(vt))] \
            if b is not None else [None] * 3
        dd_ = [gexec.to_device(x) for x in (d.indptr.astype(it), d.indices.astype(it), d.data.astype(vt))] \
            if d is not None else [None] * 3
        n = a.shape[0]
        off = gexec.to_device(np.zeros(n + 1, np.int64))
        total = C.c_int64(0)
        call("gkoc_csr_spgemm_count_" + _isuf(it), gexec.stream, n, da[0], da[1], db[0], dd_[0], off, C.byref(total))
        t = total.value
        tr, tc, tv = (gexec.to_device(np.zeros(max(t, 1), x)) for x in (it, it, vt))
        al = gexec.to_device(np.array([alpha], vt)) if alpha is not None else None
        be = gexec.to_device(np.array([beta], vt)) if beta is not None else None
        call("gkoc_csr_spgemm_expand_" + s, gexec.stream, n, al, *da, *db, be, *dd_, off, tr, tc, tv)
        return to_csr(n, (b if b is not None else a).shape[1], t, tr, tc, tv)

    def mk(r, c, dens, seed):
        rp, ci, v = random_csr(r, c, dens, seed, it, dtype=vt, empty_rows=(1,))
        return sp.csr_matrix((v, ci, rp), shape=(r, c))
    a, b, d = mk(60, 45, 0.1, 1), mk(45, 70, 0.1, 2), mk(60, 70, 0.05, 3)
    structural = (abs(a) @ abs(b)).tocsr()            # no cancellation: the pattern of the product
    for got, want, pat in ((product(a, b, None, None, None), a @ b, structural),
                           (product(a, b, 0.7, -1.3, d), vt(0.7) * (a @ b) + vt(-1.3) * d,
                            (structural + abs(d)).tocsr())):
        pat.sort_indices()
        assert np.array_equal(got.indptr, pat.indptr) and np.array_equal(got.indices, pat.indices)
        want = want.toarray()
        assert np.max(np.abs(got.toarray() - want)) <= tol * max(1.0, np.max(np.abs(want)))
    # SpGEAM: alpha A + beta B with A's entries first
    e = mk(60, 70, 0.08, 4)
    got = product(e, None, -1.3, 0.7, d)          # "d" = the first matrix (alpha = 0.7), "a" = the second (beta)
    want = (vt(0.7) * d + vt(-1.3) * e).tocsr()
    pat = (abs(d) + abs(e)).tocsr()
    pat.sort_indices()
    assert np.array_equal(got.indptr, pat.indptr) and np.array_equal(got.indices, pat.indices)
    assert np.max(np.abs(got.toarray() - want.toarray())) <= tol
    # an empty product
    z = sp.csr_matrix((5, 5), dtype=vt)
    got = product(z, z, None, None, None)
    assert got.nnz == 0 and got.indptr.tolist() == [0] * 6


@pytest.mark.parametrize("vt,it", TYPES)
def test_l1_jacobi_helpers(gexec, vt, it):
    from ginkgo_amd._lib import call
    s = _suf(vt, it)
    rows = 50
    rp, ci, v = random_csr(rows, rows, 0.12, 9, it, dtype=vt, empty_rows=(7,))
    a = sp.csr_matrix((v, ci, rp), shape=(rows, rows))
    drp, dci, dv = (gexec.to_device(x) for x in (rp, ci, v))
    da = a.toarray()
    # scalar_l1: diag += sum of |off-diagonal| in storage order
    diag0 = np.random.default_rng(1).uniform(1, 2, rows).astype(vt)
    dg = gexec.to_device(diag0.copy())
    call("gkoc_jacobi_scalar_l1_" + s, gexec.stream, rows, drp, dci, dv, dg)
    want = diag0.copy()
    for r in range(rows):
        off = vt(0)
        for k in range(rp[r], rp[r + 1]):
            if ci[k] != r:
                off = vt(off + abs(v[k]))
        want[r] = vt(want[r] + off)
    assert np.array_equal(dg.cpu().numpy(), want)
    # add_diagonal_elements
    shift = gexec.to_device(np.zeros(rows + 1, it))
    miss = C.c_int64(0)
    call("gkoc_csr_missing_diagonal_shift_" + _isuf(it), gexec.stream, rows, rows, drp, dci, shift, C.byref(miss))
    lacking = [r for r in range(rows) if r not in ci[rp[r]:rp[r + 1]]]
    assert miss.value == len(lacking) and len(lacking) > 0
    nn = len(ci) + miss.value
    nrp, nci, nv = gexec.to_device(np.zeros(rows + 1, it)), gexec.to_device(np.zeros(nn, it)), \
        gexec.to_device(np.full(nn, 9, vt))
    call("gkoc_csr_add_diagonal_fill_" + s, gexec.stream, rows, drp, dci, dv, shift, nrp, nci, nv)
    b = sp.csr_matrix((nv.cpu().numpy(), nci.cpu().numpy(), nrp.cpu().numpy()), shape=(rows, rows))
    assert np.array_equal(b.toarray(), da)
    bi, bp = nci.cpu().numpy(), nrp.cpu().numpy()
    for r in range(rows):
        cols_r = bi[bp[r]:bp[r + 1]]
        assert (cols_r == r).sum() == 1 and np.all(np.diff(cols_r) > 0)          # sorted input stays sorted
    # block_l1 with blocks of 4 rows
    nb = -(-rows // 4)
    bptr = np.minimum(np.arange(nb + 1) * 4, rows).astype(it)
    vals0 = nv.cpu().numpy().copy()
    call("gkoc_jacobi_block_l1_" + s, gexec.stream, nb, gexec.to_device(bptr), nrp, nci, nv)
    want = vals0.copy()
    for r in range(rows):
        lo, hi = (r // 4) * 4, min((r // 4) * 4 + 4, rows)
        off, at = vt(0), -1
        for k in range(bp[r], bp[r + 1]):
            if lo <= bi[k] < hi:
                if bi[k] == r:
                    at = k
                continue
            off = vt(off + abs(vals0[k]))
        want[at] = vt(want[at] + off)
    assert np.array_equal(nv.cpu().numpy(), want)
