"""Inputs of the configurations that are not stencils (BASELINE configs[4]: SuiteSparse
Janna/Flan_1565, SELL-P vs CSR, 1 and N ranks):

* read_mtx / write_mtx   MatrixMarket files (core/base/mtx_io.cpp: coordinate | array,
                         real | integer | pattern, general | symmetric | skew-symmetric;
                         1-based indices, symmetric files store one triangle)
* flan_like_rows         the stand-in used when the file is not at hand (there is no network on
                         the build boxes): A = L27(g) (x) B3 - the 27-point Laplacian's sparsity with
                         a 3 x 3 SPD block per entry = three degrees of freedom per node as in a 3-D
                         elasticity mesh; g = 80: n = 1 536 000, 123 M nonzeros, up to 81 per row
                         (Flan_1565: n = 1 564 794, 114 M).  Rows of any range, so that a rank builds
                         its own rows only.
* partition_by_nnz       contiguous row ranges with equal shares of the stored entries
                         (Partition::build_from_contiguous, include/ginkgo/core/distributed/
                         partition.hpp:262 ff.; the reference's benchmark balances rows, an FE matrix
                         with boundary rows of a third of the interior's length wants entries)

Host-side set-up (numpy / scipy); the device part starts at Csr.from_arrays / DeviceMatrixData.
"""
import io
import os

import numpy as np


class MtxError(ValueError):
    pass


def _open_text(path):
    if str(path).endswith(".gz"):
        import gzip
        return gzip.open(path, "rt")
    return open(path, "rt")


def read_mtx(path_or_file):
    """-> (n_rows, n_cols, rows, cols, vals): 0-based int64 indices, float64 values, every entry the
    file MEANS (a symmetric file's second triangle included), in file order; duplicates are kept
    (device_matrix_data::sum_duplicates is the caller's business, as in mtx_io.cpp)."""
    f = _open_text(path_or_file) if isinstance(path_or_file, (str, os.PathLike)) else path_or_file
    with f:
        header = f.readline().split()
        if len(header) < 5 or header[0] != "%%MatrixMarket" or header[1].lower() != "matrix":
            raise MtxError("not a MatrixMarket matrix file")
        layout, field, sym = (h.lower() for h in header[2:5])
        if layout not in ("coordinate", "array"):
            raise MtxError(f"unknown layout {layout}")
        if field not in ("real", "integer", "pattern", "double"):
            raise MtxError(f"unsupported field {field} (complex files: not on this path)")
        if sym not in ("general", "symmetric", "skew-symmetric"):
            raise MtxError(f"unsupported symmetry {sym}")
        line = f.readline()
        while line and (line.startswith("%") or not line.strip()):
            line = f.readline()
        dims = line.split()
        body = f.read()
    if layout == "array":
        n_rows, n_cols = int(dims[0]), int(dims[1])
        data = np.array(body.split(), dtype=np.float64)
        if sym == "general":
            if data.size != n_rows * n_cols:
                raise MtxError("array file: wrong number of values")
            dense = data.reshape(n_cols, n_rows).T          # column major
        else:
            # one triangle, column by column (the diagonal is absent in skew-symmetric files)
            dense = np.zeros((n_rows, n_cols))
            k = 0
            for j in range(n_cols):
                i0 = j if sym == "symmetric" else j + 1
                cnt = n_rows - i0
                dense[i0:, j] = data[k:k + cnt]
                k += cnt
            low = np.tril(dense, -1)
            dense = dense + (low.T if sym == "symmetric" else -low.T)
        r, c = np.nonzero(np.ones_like(dense))
        return n_rows, n_cols, r.astype(np.int64), c.astype(np.int64), dense[r, c]
    n_rows, n_cols, nnz = int(dims[0]), int(dims[1]), int(dims[2])
    width = 2 if field == "pattern" else 3
    if nnz == 0:
        z = np.zeros(0, np.int64)
        return n_rows, n_cols, z, z.copy(), np.zeros(0)
    try:
        import pandas as pd
        tab = pd.read_csv(io.StringIO(body), sep=r"\s+", header=None, comment="%", engine="c",
                          float_precision="round_trip",   # (the default parser is off by an ulp now and then)
                          dtype=np.float64 if width == 3 else np.int64).to_numpy()
    except ImportError:
        tab = np.array(body.split(), dtype=np.float64).reshape(-1, width)
    if tab.shape != (nnz, width):
        raise MtxError(f"coordinate file: {tab.shape[0]} entries of {tab.shape[1]} fields, header says {nnz} "
                       f"of {width}")
    rows = tab[:, 0].astype(np.int64) - 1
    cols = tab[:, 1].astype(np.int64) - 1
    vals = np.ones(nnz) if field == "pattern" else tab[:, 2].astype(np.float64)
    if rows.min() < 0 or rows.max() >= n_rows or cols.min() < 0 or cols.max() >= n_cols:
        raise MtxError("index out of range")
    if sym != "general":
        off = rows != cols
        if sym == "skew-symmetric" and not off.all():
            raise MtxError("skew-symmetric file with a diagonal entry")
        rows, cols, vals = (np.concatenate([rows, cols[off]]), np.concatenate([cols, rows[off]]),
                            np.concatenate([vals, vals[off] if sym == "symmetric" else -vals[off]]))
    return n_rows, n_cols, rows, cols, vals


def write_mtx(path, a, symmetric=False):
    """scipy sparse matrix -> coordinate real file (general, or the lower triangle of a symmetric one)"""
    import scipy.sparse as sp
    m = sp.coo_matrix(sp.tril(a) if symmetric else a)
    with open(path, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate real {'symmetric' if symmetric else 'general'}\n")
        f.write("% written by ginkgo_amd.workloads.write_mtx\n")
        f.write(f"{m.shape[0]} {m.shape[1]} {m.nnz}\n")
        for i, j, v in zip(m.row, m.col, m.data):
            f.write(f"{i + 1} {j + 1} {float(v)!r}\n")


def csr_from_triplets(n_rows, n_cols, rows, cols, vals, index_dtype=np.int32):
    """sorted CSR with duplicates summed (host; what Csr::read(matrix_data) does)"""
    import scipy.sparse as sp
    a = sp.coo_matrix((vals, (rows, cols)), shape=(n_rows, n_cols)).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    return a.indptr.astype(index_dtype), a.indices.astype(index_dtype), a.data.astype(np.float64)


B3 = np.array([[4.0, 1.0, 0.5], [1.0, 3.0, 0.25], [0.5, 0.25, 2.0]])


def flan_like_dims(grid):
    n = 3 * grid ** 3
    return n, 9 * (3 * grid - 2) ** 3


def flan_like_rows(grid, lo=0, hi=None, index_dtype=np.int32):
    """rows [lo, hi) of A = L27(grid) (x) B3 as CSR with GLOBAL columns:
    (row_ptrs, col_idxs, values); L27 = the 27-point stencil with 26 on the diagonal and -1 elsewhere
    (benchmark/utils/stencil_matrix.hpp:425-453), unknown 3 * node + dof"""
    n = 3 * grid ** 3
    hi = n if hi is None else hi
    node_lo, node_hi = lo // 3, -(-hi // 3)
    idx = np.arange(node_lo, node_hi, dtype=np.int64)
    ix, iy, iz = idx % grid, (idx // grid) % grid, idx // (grid * grid)
    cols_l, vals_l, mask_l = [], [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok = ((ix + dx >= 0) & (ix + dx < grid) & (iy + dy >= 0) & (iy + dy < grid) &
                      (iz + dz >= 0) & (iz + dz < grid))
                cols_l.append(idx + dx + grid * (dy + grid * dz))
                vals_l.append(np.full(idx.size, 26.0 if (dx, dy, dz) == (0, 0, 0) else -1.0))
                mask_l.append(ok)
    node_cols = np.stack(cols_l, axis=1)          # (nodes, 27), ascending per row
    node_vals = np.stack(vals_l, axis=1)
    mask = np.stack(mask_l, axis=1)
    # every node row becomes three rows of 3 * (its entries)
    per_node = mask.sum(axis=1)
    nc = node_cols[mask]
    nv = node_vals[mask]
    node_of = np.repeat(np.arange(idx.size), per_node)
    out_cols, out_vals, out_len = [], [], []
    for d in range(3):
        c = (3 * nc[:, None] + np.arange(3)[None, :]).reshape(-1)
        v = (nv[:, None] * B3[d][None, :]).reshape(-1)
        out_cols.append(c)
        out_vals.append(v)
        out_len.append(3 * per_node)
    # interleave the three dof rows of every node: row = 3 * node + d
    lens = np.stack(out_len, axis=1).reshape(-1)                    # (nodes * 3,)
    ptr = np.concatenate([[0], np.cumsum(lens)])
    cols = np.empty(ptr[-1], dtype=np.int64)
    vals = np.empty(ptr[-1], dtype=np.float64)
    starts3 = np.concatenate([[0], np.cumsum(3 * per_node)])[:-1]   # per node, within one dof's list
    for d in range(3):
        # destination offset of node k's dof-d row
        dst0 = ptr[3 * np.arange(idx.size) + d]
        rep = 3 * per_node
        dst = np.repeat(dst0 - starts3, rep) + np.arange(out_cols[d].size)
        cols[dst] = out_cols[d]
        vals[dst] = out_vals[d]
    del node_of
    r0, r1 = lo - 3 * node_lo, hi - 3 * node_lo
    k0, k1 = ptr[r0], ptr[r1]
    row_ptrs = (ptr[r0:r1 + 1] - k0).astype(index_dtype)
    return row_ptrs, cols[k0:k1].astype(index_dtype), vals[k0:k1]


def partition_by_nnz(row_nnz_prefix, n_parts, align=1):
    """offsets (n_parts + 1) of contiguous row ranges with equal shares of the stored entries;
    row_nnz_prefix: row_ptrs of the global matrix (or any non-decreasing prefix sum over rows);
    boundaries are multiples of `align` (blocks of a block preconditioner must not straddle ranks)"""
    rp = np.asarray(row_nnz_prefix, dtype=np.int64)
    n = rp.size - 1
    total = rp[-1] - rp[0]
    offsets = [0]
    for p in range(1, n_parts):
        target = rp[0] + total * p // n_parts
        r = int(np.searchsorted(rp, target))
        r = min(n, max(offsets[-1], (r + align // 2) // align * align))
        offsets.append(r)
    offsets.append(n)
    return offsets


def flan_like_row_prefix(grid):
    """row_ptrs of the stand-in without building it (for partition_by_nnz)"""
    g = np.arange(grid)
    one = 3 - (g == 0) - (g == grid - 1)                 # neighbours along one axis, incl. itself
    per_node = (one[None, None, :] * one[None, :, None] * one[:, None, None]).reshape(-1)   # z, y, x
    lens = np.repeat(3 * per_node, 3)
    return np.concatenate([[0], np.cumsum(lens)])


# ---- a second stand-in for configs[4], IRREGULAR on purpose -------------------
# flan_like_rows has 24 - 81 entries in every row; "irregular nnz/row" is what BASELINE names.  This one
# has a heavy tail: a symmetric positive definite matrix (strictly diagonally dominant) whose row lengths
# follow a power law - most rows a dozen entries, one in a thousand some hundreds - plus a few HUB rows with
# 10^3 .. 10^5 entries, beyond GKOC_CSR_LONG_ROW (4096), where the CSR kernel sums a row with the whole
# wave and SELL-P (slice size 64, padded to the longest row of the slice) stores a multiple of the entries.
# Everything is a pure function of the indices, so a rank builds its own rows only:
#   * row a is linked to a + S[k] for k < d(a), d(a) = min(96, floor(2 / u(a)^0.7)), u(a) a hash of a in
#     (0, 1]  (P(d >= k) ~ k^-1.43) - and, by symmetry, row b to b - S[k] wherever k < d(b - S[k]);
#   * hub j (row H_j) is linked to every row r with (r + 131 j) mod M_j == 0, M_j = 16 .. 6144;
#   * weights w(a, b) in [0.5, 1.5) from a hash of (min, max); a_rr = 1 + sum of the row's weights.
IRR_KMAX = 96
IRR_HUBS = 24


def _irr_offsets():
    k = np.arange(IRR_KMAX, dtype=np.int64)
    return 1 + 37 * k * k + 1009 * k


def _irr_degree(idx):
    h = (idx.astype(np.uint64) * np.uint64(2654435761) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    u = ((h >> np.uint64(8)).astype(np.float64) + 1.0) / float(1 << 24)
    return np.minimum(IRR_KMAX, np.floor(2.0 / u ** 0.7)).astype(np.int64)


def _irr_weight(a, b):
    lo, hi = np.minimum(a, b).astype(np.uint64), np.maximum(a, b).astype(np.uint64)
    h = (lo * np.uint64(2654435761) + hi * np.uint64(40503)) & np.uint64(0x3FF)
    return 0.5 + h.astype(np.float64) / 1024.0


def _irr_hubs(n):
    j = np.arange(IRR_HUBS, dtype=np.int64)
    rows = (j + 1) * n // (IRR_HUBS + 1)
    mods = 16 * (1 << (j % 8)) * (1 + j // 8)
    return rows, mods


def irregular_rows(n, lo=0, hi=None, index_dtype=np.int32):
    """rows [lo, hi) of the heavy-tailed stand-in of order n as sorted CSR with GLOBAL columns"""
    import scipy.sparse as sp
    hi = n if hi is None else hi
    r = np.arange(lo, hi, dtype=np.int64)
    s = _irr_offsets()
    d_own = _irr_degree(r)
    rows_l, cols_l = [], []
    for k in range(IRR_KMAX):
        up = (k < d_own) & (r + s[k] < n)                   # r -> r + s_k
        rows_l.append(r[up])
        cols_l.append(r[up] + s[k])
        below = r - s[k]
        ok = below >= 0
        dn = np.zeros(r.size, dtype=bool)
        dn[ok] = k < _irr_degree(below[ok])                   # (r - s_k) -> r, seen from r
        rows_l.append(r[dn])
        cols_l.append(below[dn])
    hub_rows, hub_mods = _irr_hubs(n)
    for j in range(IRR_HUBS):
        hj, mj = int(hub_rows[j]), int(hub_mods[j])
        mine = r[((r + 131 * j) % mj == 0) & (r != hj)]       # ordinary rows of the range linked to hub j
        rows_l.append(mine)
        cols_l.append(np.full(mine.size, hj, dtype=np.int64))
        if lo <= hj < hi:                                     # the hub's own row: all its spokes
            first = (-131 * j) % mj
            spokes = np.arange(first, n, mj, dtype=np.int64)
            spokes = spokes[spokes != hj]
            rows_l.append(np.full(spokes.size, hj, dtype=np.int64))
            cols_l.append(spokes)
    rows = np.concatenate(rows_l)
    cols = np.concatenate(cols_l)
    # a link reached twice (an offset that is also a spoke) is one link
    key = np.unique(rows * np.int64(n) + cols)
    rows, cols = key // n, key % n
    w = -_irr_weight(rows, cols)
    a = sp.csr_matrix((w, (rows - lo, cols)), shape=(hi - lo, n))
    diag = 1.0 - np.asarray(a.sum(axis=1)).ravel()
    a = (a + sp.csr_matrix((diag, (np.arange(hi - lo), r)), shape=(hi - lo, n))).tocsr()
    a.sort_indices()
    return a.indptr.astype(index_dtype), a.indices.astype(index_dtype), a.data.astype(np.float64)


def irregular_row_prefix(n):
    """row_ptrs of the whole stand-in without its entries (for partition_by_nnz); exact: the same links,
    counted"""
    r = np.arange(n, dtype=np.int64)
    s = _irr_offsets()
    d = _irr_degree(r)
    keys = []
    for k in range(IRR_KMAX):
        up = (k < d) & (r + s[k] < n)
        keys.append(r[up] * np.int64(n) + r[up] + s[k])
        keys.append((r[up] + s[k]) * np.int64(n) + r[up])
    hub_rows, hub_mods = _irr_hubs(n)
    for j in range(IRR_HUBS):
        hj, mj = int(hub_rows[j]), int(hub_mods[j])
        spokes = np.arange((-131 * j) % mj, n, mj, dtype=np.int64)
        spokes = spokes[spokes != hj]
        keys.append(spokes * np.int64(n) + hj)
        keys.append(hj * np.int64(n) + spokes)
    key = np.unique(np.concatenate(keys))
    lens = np.bincount(key // n, minlength=n) + 1             # + the diagonal
    return np.concatenate([[0], np.cumsum(lens)])
