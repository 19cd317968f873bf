"""Inputs of configs[4] (ginkgo_amd/workloads.py): the MatrixMarket reader against scipy's on every
header variant the reference reads (core/base/mtx_io.cpp), the Flan_1565 stand-in against its
definition kron(L27, B3), the entry-balanced contiguous partition."""
import io
import os

import numpy as np
import pytest
import scipy.io
import scipy.sparse as sp

from ginkgo_amd import workloads as w


def _dense(n_rows, n_cols, r, c, v):
    return sp.coo_matrix((v, (r, c)), shape=(n_rows, n_cols)).toarray()


@pytest.mark.parametrize("text", [
    "%%MatrixMarket matrix coordinate real general\n% c\n3 4 4\n1 1 1.5\n3 4 -2\n2 2 3e0\n1 1 0.5\n",
    "%%MatrixMarket matrix coordinate real symmetric\n3 3 4\n1 1 2\n2 1 -1\n3 2 -1\n3 3 2\n",
    "%%MatrixMarket matrix coordinate integer skew-symmetric\n3 3 2\n2 1 4\n3 1 -7\n",
    "%%MatrixMarket matrix coordinate pattern general\n2 3 3\n1 1\n2 3\n1 2\n",
    "%%MatrixMarket matrix coordinate pattern symmetric\n3 3 3\n1 1\n3 1\n2 2\n",
    "%%MatrixMarket matrix array real general\n2 3\n1\n2\n3\n4\n5\n6\n",
    "%%MatrixMarket matrix array real symmetric\n3 3\n1\n2\n3\n4\n5\n6\n",
])
def test_read_mtx_matches_scipy(text, tmp_path):
    p = tmp_path / "a.mtx"
    p.write_text(text)
    want = scipy.io.mmread(str(p))
    want = want.toarray() if sp.issparse(want) else np.asarray(want)
    got = _dense(*w.read_mtx(str(p)))
    assert np.array_equal(got, want)
    got2 = _dense(*w.read_mtx(io.StringIO(text)))
    assert np.array_equal(got2, want)


def test_read_mtx_rejects_what_it_does_not_read(tmp_path):
    for text in ("%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1 0\n",
                 "%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n",
                 "%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n",
                 "hello\n"):
        with pytest.raises(ValueError):
            w.read_mtx(io.StringIO(text))


def test_write_then_read_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    a = sp.random(40, 40, density=0.1, random_state=rng, format="csr")
    a = a + a.T + sp.eye(40)
    for sym in (False, True):
        p = tmp_path / f"r{sym}.mtx"
        w.write_mtx(str(p), a, symmetric=sym)
        rp, ci, v = w.csr_from_triplets(*w.read_mtx(str(p)))
        b = sp.csr_matrix((v, ci, rp), shape=a.shape)
        assert abs(b - a).max() == 0.0


@pytest.mark.parametrize("grid", [3, 5])
def test_flan_like_rows_is_kron_of_the_stencil_and_b3(grid, oracle):
    rp, ci, v = oracle.stencil_csr(3, grid)
    l27 = sp.csr_matrix((v, ci, rp), shape=(grid ** 3, grid ** 3))
    a = sp.kron(l27, sp.csr_matrix(w.B3), format="csr")
    a.sort_indices()
    n, nnz = w.flan_like_dims(grid)
    assert a.shape[0] == n and a.nnz == nnz
    frp, fci, fv = w.flan_like_rows(grid)
    assert np.array_equal(frp, a.indptr) and np.array_equal(fci, a.indices) and np.array_equal(fv, a.data)
    assert np.array_equal(w.flan_like_row_prefix(grid), a.indptr)
    # any row range, also one that starts and ends inside a node's three rows
    for lo, hi in ((0, 7), (4, 4), (5, n), (10, 38), (n - 2, n)):
        prp, pci, pv = w.flan_like_rows(grid, lo, hi)
        assert np.array_equal(prp, a.indptr[lo:hi + 1] - a.indptr[lo])
        assert np.array_equal(pci, a.indices[a.indptr[lo]:a.indptr[hi]])
        assert np.array_equal(pv, a.data[a.indptr[lo]:a.indptr[hi]])


def test_partition_by_nnz_balances_entries_and_respects_blocks():
    prefix = w.flan_like_row_prefix(12)
    n = prefix.size - 1
    for parts in (1, 2, 3, 8):
        off = w.partition_by_nnz(prefix, parts, align=3)
        assert off[0] == 0 and off[-1] == n and len(off) == parts + 1
        assert all(b >= a for a, b in zip(off, off[1:])) and all(o % 3 == 0 for o in off)
        share = np.diff(prefix[off])
        assert share.max() <= 1.05 * prefix[-1] / parts + 3 * 81
    # rows of very different length: equal ROW counts would give the first part 10x the entries
    lens = np.concatenate([np.full(100, 50), np.full(900, 5)])
    pre = np.concatenate([[0], np.cumsum(lens)])
    off = w.partition_by_nnz(pre, 2)
    assert abs((pre[off[1]] - pre[0]) - pre[-1] / 2) <= 50
    # more parts than rows: empty parts at the end, never a decreasing offset
    off = w.partition_by_nnz(np.array([0, 3, 6]), 5)
    assert off[0] == 0 and off[-1] == 2 and all(b >= a for a, b in zip(off, off[1:]))


def test_irregular_stand_in_is_what_it_says():
    """workloads.irregular_rows: symmetric, strictly diagonally dominant, heavy-tailed row lengths with hub rows;
    any row range equals the same rows of the whole matrix (a rank builds its own rows only); the row-pointer
    prefix used for the entry-balanced partition is exact"""
    import numpy as np
    import scipy.sparse as sp
    from ginkgo_amd import workloads as wl
    n = 30000
    rp, ci, v = wl.irregular_rows(n)
    a = sp.csr_matrix((v, ci, rp), shape=(n, n))
    assert abs(a - a.T).max() == 0.0
    off = abs(a).sum(axis=1).A1 - a.diagonal()
    assert np.all(a.diagonal() == 1.0 + off) or np.allclose(a.diagonal(), 1.0 + off, rtol=1e-15)
    lens = np.diff(rp)
    assert np.median(lens) <= 12 and lens.max() > 1000 and np.percentile(lens, 99) < 60
    assert np.all(np.diff(ci.astype(np.int64))[np.setdiff1d(np.arange(ci.size - 1), rp[1:-1] - 1)] > 0)   # sorted rows
    assert np.array_equal(wl.irregular_row_prefix(n), rp.astype(np.int64))
    hub = int(wl._irr_hubs(n)[0][0])
    for lo, hi in ((0, 17), (12345, 12400), (hub - 2, hub + 3), (n - 9, n)):
        r2, c2, v2 = wl.irregular_rows(n, lo, hi)
        assert np.array_equal(r2, rp[lo:hi + 1] - rp[lo])
        assert np.array_equal(c2, ci[rp[lo]:rp[hi]]) and np.array_equal(v2, v[rp[lo]:rp[hi]])
    offs = wl.partition_by_nnz(rp, 8, align=4)
    shares = np.diff(rp[offs].astype(np.int64))
    assert offs[0] == 0 and offs[-1] == n and all(o % 4 == 0 for o in offs[:-1])
    assert shares.max() <= 1.3 * rp[-1] / 8
