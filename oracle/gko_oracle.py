"""TEST INFRASTRUCTURE -- ctypes/numpy front-end of oracle/libgko_oracle.so.

The oracle is the CPU restatement of the reference's algorithms for the hot
path (see oracle/gko_oracle.c).  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py import this module; the product never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_VT = {np.dtype(np.float64): "f64", np.dtype(np.float32): "f32"}
_IT = {np.dtype(np.int32): "i32", np.dtype(np.int64): "i64"}


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgko_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i64(v):
    return C.c_int64(int(v))


def _val(dt, v):
    return C.c_double(float(v)) if np.dtype(dt) == np.float64 else C.c_float(float(v))


def _suf(vals, idx=None):
    s = _VT[vals.dtype]
    if idx is not None:
        s += "_" + _IT[idx.dtype]
    return s


def _as2d(x):
    x = np.asarray(x)
    return x.reshape(-1, 1) if x.ndim == 1 else x


# ------------------------------------------------------------------ SpMV
def csr_spmv(row_ptrs, cols, vals, b, alpha=None, beta=None, c=None):
    """c = A b  (alpha, beta None) or c = alpha A b + beta c."""
    b2 = np.ascontiguousarray(_as2d(b))
    n = len(row_ptrs) - 1
    nrhs = b2.shape[1]
    suf = _suf(vals, cols)
    if alpha is None:
        out = np.empty((n, nrhs), dtype=vals.dtype)
        getattr(lib(), "oracle_csr_spmv_" + suf)(
            _i64(n), _p(row_ptrs), _p(cols), _p(vals), _p(b2), _i64(nrhs),
            _p(out), _i64(nrhs), _i64(nrhs))
    else:
        out = np.array(_as2d(c), dtype=vals.dtype, order="C", copy=True)
        getattr(lib(), "oracle_csr_advanced_spmv_" + suf)(
            _i64(n), _val(vals.dtype, alpha), _p(row_ptrs), _p(cols), _p(vals),
            _p(b2), _i64(nrhs), _val(vals.dtype, beta), _p(out), _i64(nrhs),
            _i64(nrhs))
    return out if np.asarray(b).ndim == 2 else out[:, 0]


def coo_apply(mode, n_rows, rows, cols, vals, b, alpha=1.0, beta=0.0, c=None):
    """coo::spmv (mode 'spmv'), advanced_spmv, spmv2, advanced_spmv2 of the reference,
    entry by entry in storage order; c is the input/output for the modes that read it"""
    m = {"spmv": 0, "advanced_spmv": 1, "spmv2": 2, "advanced_spmv2": 3}[mode]
    b2 = np.ascontiguousarray(_as2d(b))
    nrhs = b2.shape[1]
    out = np.zeros((n_rows, nrhs), dtype=vals.dtype) if c is None else \
        np.array(_as2d(c), dtype=vals.dtype, order="C", copy=True)
    getattr(lib(), "oracle_coo_apply_" + _suf(vals, cols))(
        C.c_int(m), _i64(n_rows), _i64(len(vals)), _val(vals.dtype, alpha),
        _p(np.ascontiguousarray(rows)), _p(np.ascontiguousarray(cols)),
        _p(np.ascontiguousarray(vals)), _p(b2), _i64(nrhs), _val(vals.dtype, beta), _p(out),
        _i64(nrhs), _i64(nrhs))
    return out if np.ndim(b) == 2 else out[:, 0]


def csr_to_hybrid(row_ptrs, cols, vals, ell_lim, ell_stride=None):
    """hybrid::compute_coo_row_ptrs + csr::convert_to_hybrid.  Returns
    (ell_cols, ell_vals, coo_row_ptrs, coo_rows, coo_cols, coo_vals); the ELL part is
    column-major with leading dimension ell_stride (default n_rows)."""
    n = len(row_ptrs) - 1
    stride = n if ell_stride is None else ell_stride
    f = getattr(lib(), "oracle_csr_convert_to_hybrid_" + _suf(vals, cols))
    f.restype = C.c_int64
    crp = np.zeros(n + 1, dtype=np.int64)
    total = f(_i64(n), _p(row_ptrs), _p(cols), _p(vals), _i64(ell_lim), _i64(stride), None, None,
              _p(crp), None, None, None)
    ec = np.full(stride * ell_lim, -7, dtype=cols.dtype)
    ev = np.full(stride * ell_lim, np.nan, dtype=vals.dtype)
    cr = np.full(total, -7, dtype=cols.dtype)
    cc = np.full(total, -7, dtype=cols.dtype)
    cv = np.full(total, np.nan, dtype=vals.dtype)
    f(_i64(n), _p(row_ptrs), _p(cols), _p(vals), _i64(ell_lim), _i64(stride), _p(ec), _p(ev),
      _p(crp), _p(cr), _p(cc), _p(cv))
    return ec, ev, crp, cr, cc, cv


def csr_transpose(n_rows, n_cols, row_ptrs, cols, vals):
    """csr::transpose of the reference: (t_row_ptrs, t_cols, t_vals)"""
    trp = np.zeros(n_cols + 1, dtype=cols.dtype)
    tc = np.full(len(vals), -7, dtype=cols.dtype)
    tv = np.full(len(vals), np.nan, dtype=vals.dtype)
    getattr(lib(), "oracle_csr_transpose_" + _suf(vals, cols))(
        _i64(n_rows), _i64(n_cols), _p(np.ascontiguousarray(row_ptrs)), _p(np.ascontiguousarray(cols)),
        _p(np.ascontiguousarray(vals)), _p(trp), _p(tc), _p(tv))
    return trp, tc, tv


def md_sort_row_major(rows, cols, vals):
    """components::sort_row_major of the reference: sorted copies (rows, cols, vals)"""
    r, c, v = (np.array(a, copy=True, order="C") for a in (rows, cols, vals))
    getattr(lib(), "oracle_md_sort_row_major_" + _suf(v, c))(_i64(len(v)), _p(r), _p(c), _p(v))
    return r, c, v


def _md_compact(name, rows, cols, vals):
    r, c, v = (np.ascontiguousarray(a) for a in (rows, cols, vals))
    orow, ocol, oval = np.empty_like(r), np.empty_like(c), np.empty_like(v)
    fn = getattr(lib(), name + _suf(v, c))
    fn.restype = C.c_int64
    n = fn(_i64(len(v)), _p(r), _p(c), _p(v), _p(orow), _p(ocol), _p(oval))
    return orow[:n].copy(), ocol[:n].copy(), oval[:n].copy()


def md_remove_zeros(rows, cols, vals):
    """components::remove_zeros of the reference: the kept entries, in order"""
    return _md_compact("oracle_md_remove_zeros_", rows, cols, vals)


def md_sum_duplicates(rows, cols, vals):
    """components::sum_duplicates of the reference on row-major sorted input"""
    return _md_compact("oracle_md_sum_duplicates_", rows, cols, vals)


def ell_spmv(n_rows, k, stride, cols, vals, b, alpha=None, beta=None, c=None):
    b2 = np.ascontiguousarray(_as2d(b))
    nrhs = b2.shape[1]
    suf = _suf(vals, cols)
    if alpha is None:
        out = np.empty((n_rows, nrhs), dtype=vals.dtype)
        getattr(lib(), "oracle_ell_spmv_" + suf)(
            _i64(n_rows), _i64(k), _i64(stride), _p(cols), _p(vals), _p(b2),
            _i64(nrhs), _p(out), _i64(nrhs), _i64(nrhs))
    else:
        out = np.array(_as2d(c), dtype=vals.dtype, order="C", copy=True)
        getattr(lib(), "oracle_ell_advanced_spmv_" + suf)(
            _i64(n_rows), _i64(k), _i64(stride), _val(vals.dtype, alpha),
            _p(cols), _p(vals), _p(b2), _i64(nrhs), _val(vals.dtype, beta),
            _p(out), _i64(nrhs), _i64(nrhs))
    return out if np.asarray(b).ndim == 2 else out[:, 0]


def _mixed_suf(vals, b, out_dtype, cols):
    out_dtype = np.dtype(out_dtype)
    if vals.dtype == b.dtype == out_dtype:
        raise ValueError("uniform triple: use csr_spmv / ell_spmv")
    return "_".join((_VT[vals.dtype], _VT[b.dtype], _VT[out_dtype], _IT[cols.dtype]))


def csr_spmv_mixed(row_ptrs, cols, vals, b, out_dtype, alpha=None, beta=None, c=None):
    """csr::spmv<MatrixValueType, InputValueType, OutputValueType> of a GINKGO_MIXED_PRECISION
    core: vals / b / the result each float32 or float64, not all the same; alpha is rounded to
    the matrix' type, beta to the output's (they are Dense<MatrixValueType> / Dense<OutputValueType>)"""
    b2 = np.ascontiguousarray(_as2d(b))
    n = len(row_ptrs) - 1
    nrhs = b2.shape[1]
    suf = _mixed_suf(vals, b2, out_dtype, cols)
    if alpha is None:
        out = np.empty((n, nrhs), dtype=out_dtype)
        getattr(lib(), "oracle_csr_spmv_mixed_" + suf)(
            _i64(n), _p(row_ptrs), _p(cols), _p(vals), _p(b2), _i64(nrhs), _p(out), _i64(nrhs), _i64(nrhs))
    else:
        out = np.array(_as2d(c), dtype=out_dtype, order="C", copy=True)
        getattr(lib(), "oracle_csr_advanced_spmv_mixed_" + suf)(
            _i64(n), _val(vals.dtype, alpha), _p(row_ptrs), _p(cols), _p(vals), _p(b2), _i64(nrhs),
            _val(out_dtype, beta), _p(out), _i64(nrhs), _i64(nrhs))
    return out if np.asarray(b).ndim == 2 else out[:, 0]


def ell_spmv_mixed(n_rows, k, stride, cols, vals, b, out_dtype, alpha=None, beta=None, c=None):
    """ell::spmv / advanced_spmv for a non-uniform (matrix, input, output) triple"""
    b2 = np.ascontiguousarray(_as2d(b))
    nrhs = b2.shape[1]
    suf = _mixed_suf(vals, b2, out_dtype, cols)
    if alpha is None:
        out = np.empty((n_rows, nrhs), dtype=out_dtype)
        getattr(lib(), "oracle_ell_spmv_mixed_" + suf)(
            _i64(n_rows), _i64(k), _i64(stride), _p(cols), _p(vals), _p(b2), _i64(nrhs), _p(out),
            _i64(nrhs), _i64(nrhs))
    else:
        out = np.array(_as2d(c), dtype=out_dtype, order="C", copy=True)
        getattr(lib(), "oracle_ell_advanced_spmv_mixed_" + suf)(
            _i64(n_rows), _i64(k), _i64(stride), _val(vals.dtype, alpha), _p(cols), _p(vals), _p(b2),
            _i64(nrhs), _val(out_dtype, beta), _p(out), _i64(nrhs), _i64(nrhs))
    return out if np.asarray(b).ndim == 2 else out[:, 0]


def sellp_spmv(n_rows, slice_size, slice_sets, slice_lengths, cols, vals, b,
               alpha=None, beta=None, c=None):
    b2 = np.ascontiguousarray(_as2d(b))
    nrhs = b2.shape[1]
    suf = _suf(vals, cols)
    if alpha is None:
        out = np.empty((n_rows, nrhs), dtype=vals.dtype)
        getattr(lib(), "oracle_sellp_spmv_" + suf)(
            _i64(n_rows), _i64(slice_size), _p(slice_sets), _p(slice_lengths),
            _p(cols), _p(vals), _p(b2), _i64(nrhs), _p(out), _i64(nrhs),
            _i64(nrhs))
    else:
        out = np.array(_as2d(c), dtype=vals.dtype, order="C", copy=True)
        getattr(lib(), "oracle_sellp_advanced_spmv_" + suf)(
            _i64(n_rows), _i64(slice_size), _val(vals.dtype, alpha),
            _p(slice_sets), _p(slice_lengths), _p(cols), _p(vals), _p(b2),
            _i64(nrhs), _val(vals.dtype, beta), _p(out), _i64(nrhs), _i64(nrhs))
    return out if np.asarray(b).ndim == 2 else out[:, 0]


# ----------------------------------------------------------- conversions
def sellp_compute_slice_sets(row_ptrs, slice_size=64, stride_factor=1):
    n = len(row_ptrs) - 1
    ns = (n + slice_size - 1) // slice_size
    sets = np.zeros(ns + 1, dtype=np.uint64)
    lens = np.zeros(ns, dtype=np.uint64)
    getattr(lib(), "oracle_sellp_compute_slice_sets_f64_" + _IT[row_ptrs.dtype])(
        _i64(n), _i64(slice_size), _i64(stride_factor), _p(row_ptrs), _p(sets),
        _p(lens))
    return sets, lens


def csr_to_sellp(row_ptrs, cols, vals, slice_size=64, stride_factor=1):
    n = len(row_ptrs) - 1
    sets, lens = sellp_compute_slice_sets(row_ptrs, slice_size, stride_factor)
    total = int(sets[-1]) * slice_size
    s_cols = np.full(total, -1, dtype=cols.dtype)
    s_vals = np.zeros(total, dtype=vals.dtype)
    getattr(lib(), "oracle_csr_convert_to_sellp_" + _suf(vals, cols))(
        _i64(n), _i64(slice_size), _p(row_ptrs), _p(cols), _p(vals), _p(sets),
        _p(s_cols), _p(s_vals))
    return sets, lens, s_cols, s_vals


def csr_to_ell(row_ptrs, cols, vals, k=None, stride=None):
    n = len(row_ptrs) - 1
    if k is None:
        k = int(np.max(np.diff(row_ptrs))) if n > 0 else 0
    if stride is None:
        stride = n
    e_cols = np.full(k * stride, -1, dtype=cols.dtype)
    e_vals = np.zeros(k * stride, dtype=vals.dtype)
    getattr(lib(), "oracle_csr_convert_to_ell_" + _suf(vals, cols))(
        _i64(n), _p(row_ptrs), _p(cols), _p(vals), _i64(k), _i64(stride),
        _p(e_cols), _p(e_vals))
    return k, stride, e_cols, e_vals


def csr_extract_diagonal(n_rows, n_cols, row_ptrs, cols, vals):
    d = np.zeros(min(n_rows, n_cols), dtype=vals.dtype)
    getattr(lib(), "oracle_csr_extract_diagonal_" + _suf(vals, cols))(
        _i64(n_rows), _i64(n_cols), _p(row_ptrs), _p(cols), _p(vals), _p(d))
    return d


# ------------------------------------------------------------ block-Jacobi
def jacobi_storage_scheme(max_block_size, warp_size=64):
    """include/ginkgo/core/preconditioner/jacobi.hpp:589-627
    (compute_storage_scheme with max_block_stride = warp size on HIP)."""
    p2 = 1
    while p2 < max_block_size:
        p2 *= 2
    group_size = warp_size // p2
    block_offset = max_block_size
    block_stride = group_size * block_offset
    group_offset = max_block_size * block_stride
    group_power = group_size.bit_length() - 1
    return block_offset, group_offset, group_power


def jacobi_storage_size(scheme, num_blocks):
    bo, go, gp = scheme
    gs = 1 << gp
    return ((num_blocks + gs - 1) // gs) * go


def jacobi_find_blocks(row_ptrs, cols, max_block_size):
    n = len(row_ptrs) - 1
    ptrs = np.zeros(n + 1, dtype=row_ptrs.dtype)
    f = getattr(lib(), "oracle_jacobi_find_blocks_f64_" + _IT[row_ptrs.dtype])
    f.restype = C.c_int64
    nb = f(_i64(n), _p(row_ptrs), _p(cols), C.c_uint32(max_block_size), _p(ptrs))
    return int(nb), ptrs


def jacobi_generate(row_ptrs, cols, vals, num_blocks, scheme, block_ptrs):
    bo, go, gp = scheme
    blocks = np.zeros(jacobi_storage_size(scheme, num_blocks), dtype=vals.dtype)
    getattr(lib(), "oracle_jacobi_generate_" + _suf(vals, cols))(
        _p(row_ptrs), _p(cols), _p(vals), _i64(num_blocks), _i64(bo), _i64(go),
        C.c_uint32(gp), _p(block_ptrs), _p(blocks))
    return blocks


def jacobi_apply(num_blocks, scheme, block_ptrs, blocks, b, alpha=1.0, beta=0.0,
                 x=None):
    bo, go, gp = scheme
    b2 = np.ascontiguousarray(_as2d(b))
    nrhs = b2.shape[1]
    if x is None:
        out = np.zeros_like(b2)
    else:
        out = np.array(_as2d(x), dtype=blocks.dtype, order="C", copy=True)
    getattr(lib(), "oracle_jacobi_apply_" + _suf(blocks, block_ptrs))(
        _i64(num_blocks), _i64(bo), _i64(go), C.c_uint32(gp), _p(block_ptrs),
        _p(blocks), _val(blocks.dtype, alpha), _p(b2), _i64(nrhs),
        _val(blocks.dtype, beta), _p(out), _i64(nrhs), _i64(nrhs))
    return out if np.asarray(b).ndim == 2 else out[:, 0]


def jacobi_convert_storage(num_blocks, scheme, blocks, prec):
    """blocks (float64, as jacobi_generate returns them) narrowed in place to the storage
    type of precision_reduction byte `prec`; returns the same buffer"""
    bo, go, gp = scheme
    out = np.array(blocks, dtype=np.float64, copy=True)
    lib().oracle_jacobi_convert_storage_f64(_i64(num_blocks), _i64(bo), _i64(go), C.c_uint32(gp),
                                            _p(out), C.c_int(prec))
    return out


def jacobi_apply_stored(num_blocks, scheme, block_ptrs, blocks, prec, b, alpha=1.0, beta=0.0, x=None):
    bo, go, gp = scheme
    b2 = np.ascontiguousarray(_as2d(b), dtype=np.float64)
    nrhs = b2.shape[1]
    out = np.zeros_like(b2) if x is None else np.array(_as2d(x), dtype=np.float64, order="C", copy=True)
    lib().oracle_jacobi_apply_stored_f64_i32(
        _i64(num_blocks), _i64(bo), _i64(go), C.c_uint32(gp), _p(np.ascontiguousarray(block_ptrs, np.int32)),
        _p(blocks), C.c_int(prec), C.c_double(alpha), _p(b2), _i64(nrhs), C.c_double(beta), _p(out),
        _i64(nrhs), _i64(nrhs))
    return out if np.asarray(b).ndim == 2 else out[:, 0]


def jacobi_generate_adaptive(row_ptrs, cols, vals, num_blocks, scheme, block_ptrs, accuracy=1e-1,
                             requested=None):
    """adaptive generate: (raw blocks as float64 words, chosen precision per block, conditioning).
    requested: per-block precision_reduction bytes (0xff = autodetect), default all autodetect"""
    bo, go, gp = scheme
    gs = 1 << gp
    storage = ((num_blocks + gs - 1) // gs) * go
    blocks = np.zeros(storage)
    prec = np.full(num_blocks, 0xff, np.uint8) if requested is None else \
        np.resize(np.asarray(requested, np.uint8), num_blocks).copy()
    cond = np.zeros(num_blocks)
    lib().oracle_jacobi_generate_adaptive_f64_i32(
        _p(np.ascontiguousarray(row_ptrs, np.int32)), _p(np.ascontiguousarray(cols, np.int32)),
        _p(np.ascontiguousarray(vals, np.float64)), _i64(num_blocks), _i64(bo), _i64(go), C.c_uint32(gp),
        _p(np.ascontiguousarray(block_ptrs, np.int32)), C.c_double(accuracy), _p(prec), _p(cond), _p(blocks))
    return blocks, prec, cond


def jacobi_apply_adaptive(num_blocks, scheme, block_ptrs, blocks, prec, b, alpha=1.0, beta=0.0, x=None):
    bo, go, gp = scheme
    b2 = np.ascontiguousarray(_as2d(b), dtype=np.float64)
    nrhs = b2.shape[1]
    out = np.zeros_like(b2) if x is None else np.array(_as2d(x), dtype=np.float64, order="C", copy=True)
    lib().oracle_jacobi_apply_adaptive_f64_i32(
        _i64(num_blocks), _i64(bo), _i64(go), C.c_uint32(gp), _p(np.ascontiguousarray(block_ptrs, np.int32)),
        _p(blocks), _p(np.ascontiguousarray(prec, np.uint8)), C.c_double(alpha), _p(b2), _i64(nrhs),
        C.c_double(beta), _p(out), _i64(nrhs), _i64(nrhs))
    return out if np.asarray(b).ndim == 2 else out[:, 0]


_JT = {np.dtype(np.float32): ("f32", np.float32), np.dtype(np.complex64): ("c64", np.float32),
       np.dtype(np.complex128): ("c128", np.float64)}


def jacobi_generate_adaptive_t(row_ptrs, cols, vals, num_blocks, scheme, block_ptrs, accuracy=1e-1,
                               requested=None):
    """jacobi_generate_adaptive for float32 / complex64 / complex128 values (gko_oracle_jacobi_types.inc):
    (blocks in the value type's words, precision per block, conditioning in the component type)"""
    bo, go, gp = scheme
    vals = np.ascontiguousarray(vals)
    name, rdt = _JT[vals.dtype]
    gs = 1 << gp
    blocks = np.zeros(((num_blocks + gs - 1) // gs) * go, vals.dtype)
    prec = np.full(num_blocks, 0xff, np.uint8) if requested is None else \
        np.resize(np.asarray(requested, np.uint8), num_blocks).copy()
    cond = np.zeros(num_blocks, rdt)
    getattr(lib(), f"oracle_jacobi_generate_adaptive_{name}_i32")(
        _p(np.ascontiguousarray(row_ptrs, np.int32)), _p(np.ascontiguousarray(cols, np.int32)), _p(vals),
        _i64(num_blocks), _i64(bo), _i64(go), C.c_uint32(gp), _p(np.ascontiguousarray(block_ptrs, np.int32)),
        _val(rdt, accuracy), _p(prec), _p(cond), _p(blocks))
    return blocks, prec, cond


def jacobi_apply_adaptive_t(num_blocks, scheme, block_ptrs, blocks, prec, b, alpha=None, beta=None, x=None):
    """x = M b (alpha is None) or x = alpha M b + beta x, blocks widened from their storage types"""
    bo, go, gp = scheme
    dt = blocks.dtype
    name, _ = _JT[dt]
    b2 = np.ascontiguousarray(_as2d(b), dtype=dt)
    nrhs = b2.shape[1]
    out = np.zeros_like(b2) if x is None else np.array(_as2d(x), dtype=dt, order="C", copy=True)
    a_, b_ = (None, None) if alpha is None else (np.asarray([alpha], dt), np.asarray([beta], dt))
    getattr(lib(), f"oracle_jacobi_apply_adaptive_{name}_i32")(
        _i64(num_blocks), _i64(bo), _i64(go), C.c_uint32(gp), _p(np.ascontiguousarray(block_ptrs, np.int32)),
        _p(blocks), _p(np.ascontiguousarray(prec, np.uint8)), _p(a_), _p(b2), _i64(nrhs), _p(b_), _p(out),
        _i64(nrhs), _i64(nrhs))
    return out if np.asarray(b).ndim == 2 else out[:, 0]


def jacobi_transpose_adaptive_t(num_blocks, scheme, block_ptrs, blocks, prec, conj=False):
    bo, go, gp = scheme
    name, _ = _JT[blocks.dtype]
    out = np.zeros_like(blocks)
    getattr(lib(), f"oracle_jacobi_transpose_adaptive_{name}_i32")(
        _i64(num_blocks), _i64(bo), _i64(go), C.c_uint32(gp), _p(np.ascontiguousarray(block_ptrs, np.int32)),
        _p(blocks), _p(np.ascontiguousarray(prec, np.uint8)), C.c_int(1 if conj else 0), _p(out))
    return out


def jacobi_invert_diagonal(diag):
    inv = np.empty_like(diag)
    getattr(lib(), "oracle_jacobi_invert_diagonal_" + _VT[diag.dtype])(
        _i64(len(diag)), _p(diag), _p(inv))
    return inv


def jacobi_scalar_apply(inv_diag, b, alpha=None, beta=None, x=None):
    b2 = np.ascontiguousarray(_as2d(b))
    rows, cols = b2.shape
    dt = inv_diag.dtype
    if alpha is None:
        out = np.empty_like(b2)
        getattr(lib(), "oracle_jacobi_scalar_apply_" + _VT[dt])(
            _i64(rows), _i64(cols), _p(inv_diag), C.c_int(0), _val(dt, 1),
            _p(b2), _i64(cols), _val(dt, 0), _p(out), _i64(cols))
    else:
        out = np.array(_as2d(x), dtype=dt, order="C", copy=True)
        getattr(lib(), "oracle_jacobi_scalar_apply_" + _VT[dt])(
            _i64(rows), _i64(cols), _p(inv_diag), C.c_int(1), _val(dt, alpha),
            _p(b2), _i64(cols), _val(dt, beta), _p(out), _i64(cols))
    return out if np.asarray(b).ndim == 2 else out[:, 0]


# ----------------------------------------------------------------- dense
def dense_scale(alpha, x, inverse=False):
    x2 = np.array(_as2d(x), order="C", copy=True)
    a = np.ascontiguousarray(np.atleast_1d(alpha).astype(x2.dtype))
    name = "oracle_dense_inv_scale_" if inverse else "oracle_dense_scale_"
    getattr(lib(), name + _VT[x2.dtype])(
        _i64(x2.shape[0]), _i64(x2.shape[1]), _p(a), _i64(len(a)), _p(x2),
        _i64(x2.shape[1]))
    return x2 if np.asarray(x).ndim == 2 else x2[:, 0]


def dense_add_scaled(alpha, x, y, subtract=False):
    x2 = np.ascontiguousarray(_as2d(x))
    y2 = np.array(_as2d(y), order="C", copy=True)
    a = np.ascontiguousarray(np.atleast_1d(alpha).astype(x2.dtype))
    getattr(lib(), "oracle_dense_add_scaled_" + _VT[x2.dtype])(
        _i64(x2.shape[0]), _i64(x2.shape[1]), _p(a), _i64(len(a)), _p(x2),
        _i64(x2.shape[1]), _p(y2), _i64(y2.shape[1]), C.c_int(int(subtract)))
    return y2 if np.asarray(y).ndim == 2 else y2[:, 0]


def dense_dot(x, y):
    x2, y2 = np.ascontiguousarray(_as2d(x)), np.ascontiguousarray(_as2d(y))
    res = np.zeros(x2.shape[1], dtype=x2.dtype)
    getattr(lib(), "oracle_dense_compute_dot_" + _VT[x2.dtype])(
        _i64(x2.shape[0]), _i64(x2.shape[1]), _p(x2), _i64(x2.shape[1]), _p(y2),
        _i64(y2.shape[1]), _p(res))
    return res


def dense_norm2(x, squared=False):
    x2 = np.ascontiguousarray(_as2d(x))
    res = np.zeros(x2.shape[1], dtype=x2.dtype)
    getattr(lib(), "oracle_dense_compute_norm2_" + _VT[x2.dtype])(
        _i64(x2.shape[0]), _i64(x2.shape[1]), _p(x2), _i64(x2.shape[1]), _p(res),
        C.c_int(int(squared)))
    return res


# -------------------------------------------------------------- CG steps
def cg_initialize(b):
    b2 = np.ascontiguousarray(_as2d(b))
    rows, cols = b2.shape
    r, z, p, q = (np.full_like(b2, np.nan) for _ in range(4))
    prev_rho = np.full(cols, np.nan, dtype=b2.dtype)
    rho = np.full(cols, np.nan, dtype=b2.dtype)
    stop = np.full(cols, 0xFF, dtype=np.uint8)
    getattr(lib(), "oracle_cg_initialize_" + _VT[b2.dtype])(
        _i64(rows), _i64(cols), _p(b2), _i64(cols), _p(r), _i64(cols), _p(z),
        _i64(cols), _p(p), _i64(cols), _p(q), _i64(cols), _p(prev_rho), _p(rho),
        _p(stop))
    return r, z, p, q, prev_rho, rho, stop


def cg_step_1(p, z, rho, prev_rho, stop):
    p2 = np.array(_as2d(p), order="C", copy=True)
    z2 = np.ascontiguousarray(_as2d(z))
    getattr(lib(), "oracle_cg_step_1_" + _VT[p2.dtype])(
        _i64(p2.shape[0]), _i64(p2.shape[1]), _p(p2), _i64(p2.shape[1]), _p(z2),
        _i64(z2.shape[1]), _p(np.ascontiguousarray(rho)),
        _p(np.ascontiguousarray(prev_rho)), _p(np.ascontiguousarray(stop)))
    return p2


def cg_step_2(x, r, p, q, beta, rho, stop):
    x2 = np.array(_as2d(x), order="C", copy=True)
    r2 = np.array(_as2d(r), order="C", copy=True)
    p2, q2 = np.ascontiguousarray(_as2d(p)), np.ascontiguousarray(_as2d(q))
    c = x2.shape[1]
    getattr(lib(), "oracle_cg_step_2_" + _VT[x2.dtype])(
        _i64(x2.shape[0]), _i64(c), _p(x2), _i64(c), _p(r2), _i64(c), _p(p2),
        _i64(c), _p(q2), _i64(c), _p(np.ascontiguousarray(beta)),
        _p(np.ascontiguousarray(rho)), _p(np.ascontiguousarray(stop)))
    return x2, r2


def residual_norm(tau, orig_tau, goal, stopping_id, set_finalized, stop,
                  implicit=False):
    stop2 = np.array(stop, dtype=np.uint8, copy=True)
    tau = np.ascontiguousarray(tau)
    orig_tau = np.ascontiguousarray(orig_tau)
    changed = C.c_int(0)
    f = getattr(lib(), "oracle_residual_norm_" + _VT[tau.dtype])
    f.restype = C.c_int
    allc = f(_i64(len(tau)), _p(tau), _p(orig_tau), _val(tau.dtype, goal),
             C.c_uint8(stopping_id), C.c_int(int(set_finalized)), _p(stop2),
             C.c_int(int(implicit)), C.byref(changed))
    return bool(allc), bool(changed.value), stop2


# ---------------------------------------------------------------- stencils
def stencil_subdomain(nd, dims, pos, g, restricted):
    """COO entries (global indices) of the rows owned by subdomain `pos`."""
    dims = np.asarray(dims, dtype=np.int64)
    pos = np.asarray(pos, dtype=np.int64)
    f = lib().oracle_stencil_subdomain
    f.restype = C.c_int64
    local = C.c_int64(0)
    nnz = f(C.c_int(nd), _p(dims), _p(pos), _i64(g), C.c_int(int(restricted)),
            None, None, None, C.byref(local))
    rows = np.empty(nnz, dtype=np.int64)
    cols = np.empty(nnz, dtype=np.int64)
    vals = np.empty(nnz, dtype=np.float64)
    f(C.c_int(nd), _p(dims), _p(pos), _i64(g), C.c_int(int(restricted)),
      _p(rows), _p(cols), _p(vals), C.byref(local))
    return rows, cols, vals, int(local.value)


def coo_to_csr(rows, cols, vals, row_offset, n_rows):
    row_ptrs = np.zeros(n_rows + 1, dtype=np.int64)
    out_cols = np.empty(len(rows), dtype=np.int64)
    out_vals = np.empty(len(rows), dtype=np.float64)
    lib().oracle_coo_to_csr(_i64(len(rows)), _p(rows), _p(cols), _p(vals),
                            _i64(row_offset), _i64(n_rows), _p(row_ptrs),
                            _p(out_cols), _p(out_vals))
    return row_ptrs, out_cols, out_vals


def stencil_csr(nd, g, restricted=False):
    """int32 CSR of the single-domain 5/9-pt (nd=2) or 7/27-pt (nd=3) stencil,
    benchmark/utils/stencil_matrix.hpp semantics."""
    f = lib().oracle_stencil_csr_i32
    f.restype = C.c_int64
    n = g ** nd
    nnz = f(C.c_int(nd), _i64(g), C.c_int(int(restricted)), None, None, None)
    row_ptrs = np.zeros(n + 1, dtype=np.int32)
    cols = np.empty(nnz, dtype=np.int32)
    vals = np.empty(nnz, dtype=np.float64)
    f(C.c_int(nd), _i64(g), C.c_int(int(restricted)), _p(row_ptrs), _p(cols),
      _p(vals))
    return row_ptrs, cols, vals


# ---------------------------------------------------------------- CG solve
class _Precond(C.Structure):
    _fields_ = [("precond", C.c_int), ("inv_diag", C.c_void_p),
                ("num_blocks", C.c_int64), ("block_offset", C.c_int64),
                ("group_offset", C.c_int64), ("group_power", C.c_uint32),
                ("block_ptrs", C.c_void_p), ("blocks", C.c_void_p)]


def _make_precond(row_ptrs, cols, vals, precond, max_block_size):
    """oracle_precond for precond in {None, 'scalar', 'block'} + the arrays it points to"""
    n = len(row_ptrs) - 1
    m = _Precond()
    keep = []
    if precond == "scalar":
        d = csr_extract_diagonal(n, n, row_ptrs, cols, vals)
        inv = jacobi_invert_diagonal(d)
        keep.append(inv)
        m.precond, m.inv_diag = 1, inv.ctypes.data
    elif precond == "block":
        nb, ptrs = jacobi_find_blocks(row_ptrs, cols, max_block_size)
        scheme = jacobi_storage_scheme(max_block_size)
        blocks = jacobi_generate(row_ptrs, cols, vals, nb, scheme, ptrs)
        keep += [ptrs, blocks]
        m.precond, m.num_blocks = 2, nb
        m.block_offset, m.group_offset, m.group_power = scheme
        m.block_ptrs, m.blocks = ptrs.ctypes.data, blocks.ctypes.data
    else:
        m.precond = 0
    return m, keep


KRYLOV_KINDS = {"bicgstab": 1, "cgs": 2, "fcg": 3, "pipe_cg": 4, "ir": 5, "chebyshev": 6, "bicg": 7, "gcr": 8, "minres": 9}


def krylov_solve(kind, row_ptrs, cols, vals, b, x0=None, max_iters=1000, reduction=1e-10,
                 baseline="rhs_norm", precond=None, max_block_size=8, relaxation=1.0,
                 foci=(0.0, 1.0), krylov_dim=100):
    """Bicgstab / Cgs / Fcg / PipeCg / Ir (relaxation; inner solver = precond) /
    Chebyshev (foci) with Combined(Iteration, ResidualNorm); f64 / int32, one rhs."""
    n = len(row_ptrs) - 1
    assert vals.dtype == np.float64 and cols.dtype == np.int32
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64, copy=True)
    b = np.ascontiguousarray(b, dtype=np.float64)
    m, keep = _make_precond(row_ptrs, cols, vals, precond, max_block_size)
    base = {"rhs_norm": 0, "initial_resnorm": 1, "absolute": 2}[baseline]
    resnorm = C.c_double(0)
    f = lib().oracle_krylov_solve_f64_i32
    f.restype = C.c_int64
    iters = f(C.c_int(KRYLOV_KINDS[kind]), _i64(n), _p(row_ptrs), _p(cols), _p(vals), C.byref(m),
              _p(b), _p(x), _i64(max_iters), C.c_double(reduction), C.c_int(base),
              C.c_double(relaxation if kind == "ir" else (krylov_dim if kind == "gcr" else foci[0])),
              C.c_double(foci[1]),
              C.byref(resnorm))
    del keep
    return x, int(iters), resnorm.value


def krylov_step(name, rows, cols, *arrays):
    """oracle_<name>_<f64|f32>(rows, cols, ld = cols, arrays...): the step kernels of
    bicgstab / cgs / fcg / pipe_cg, operands in the order of the reference kernel's
    signature; the arrays (C-contiguous numpy) are updated in place."""
    vt = next(a.dtype for a in arrays if a.dtype not in (np.uint8, np.uint64))
    for a in arrays:
        assert a.flags.c_contiguous and a.dtype in (vt, np.uint8, np.uint64), "mixed value types"
    dims = [_i64(cols)] if all(a.ndim == 1 for a in arrays) else [_i64(rows), _i64(cols), _i64(cols)]
    getattr(lib(), f"oracle_{name}_{_VT[np.dtype(vt)]}")(*dims, *[_p(a) for a in arrays])


def cg_solve(row_ptrs, cols, vals, b, x0=None, max_iters=1000, reduction=1e-10,
             baseline="rhs_norm", precond=None, max_block_size=8,
             return_history=False):
    """Cg(Combined(Iteration, ResidualNorm)) with precond in
    {None, 'scalar', 'block'}; f64 / int32, one right-hand side."""
    n = len(row_ptrs) - 1
    assert vals.dtype == np.float64 and cols.dtype == np.int32
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64, copy=True)
    b = np.ascontiguousarray(b, dtype=np.float64)
    m, keep = _make_precond(row_ptrs, cols, vals, precond, max_block_size)
    base = {"rhs_norm": 0, "initial_resnorm": 1, "absolute": 2}[baseline]
    resnorm = C.c_double(0)
    hist = np.full(max_iters + 1, np.nan) if return_history else None
    f = lib().oracle_cg_solve_f64_i32
    f.restype = C.c_int64
    iters = f(_i64(n), _p(row_ptrs), _p(cols), _p(vals), C.byref(m), _p(b),
              _p(x), _i64(max_iters), C.c_double(reduction), C.c_int(base),
              C.byref(resnorm), _p(hist))
    if return_history:
        return x, int(iters), resnorm.value, hist[:iters + 1]
    return x, int(iters), resnorm.value


def gmres_solve(row_ptrs, cols, vals, b, x0=None, krylov_dim=100, ortho="mgs",
                max_iters=1000, reduction=1e-10, precond=None, max_block_size=8):
    """Gmres(Combined(Iteration, ResidualNorm(rhs_norm))), non-flexible, one
    right-hand side, f64 / int32 (core/solver/gmres.cpp:321-621)."""
    n = len(row_ptrs) - 1
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64, copy=True)
    b = np.ascontiguousarray(b, dtype=np.float64)
    m = _Precond()
    keep = []
    if precond == "scalar":
        inv = jacobi_invert_diagonal(csr_extract_diagonal(n, n, row_ptrs, cols, vals))
        keep.append(inv)
        m.precond, m.inv_diag = 1, inv.ctypes.data
    elif precond == "block":
        nb, ptrs = jacobi_find_blocks(row_ptrs, cols, max_block_size)
        scheme = jacobi_storage_scheme(max_block_size)
        blocks = jacobi_generate(row_ptrs, cols, vals, nb, scheme, ptrs)
        keep += [ptrs, blocks]
        m.precond, m.num_blocks = 2, nb
        m.block_offset, m.group_offset, m.group_power = scheme
        m.block_ptrs, m.blocks = ptrs.ctypes.data, blocks.ctypes.data
    resnorm = C.c_double(0)
    f = lib().oracle_gmres_solve_f64_i32
    f.restype = C.c_int64
    iters = f(_i64(n), _p(row_ptrs), _p(cols), _p(vals), C.byref(m), _p(b), _p(x),
              _i64(krylov_dim), C.c_int({"mgs": 0, "cgs": 1, "cgs2": 2}[ortho]),
              _i64(max_iters), C.c_double(reduction), C.byref(resnorm))
    return x, int(iters), resnorm.value


# ------------------------------------------------------------ GMRES kernels
def gmres_initialize(b, krylov_dim):
    b2 = np.ascontiguousarray(_as2d(b))
    rows, cols = b2.shape
    res = np.full_like(b2, np.nan)
    gsin = np.full((krylov_dim, cols), np.nan, dtype=b2.dtype)
    gcos = np.full((krylov_dim, cols), np.nan, dtype=b2.dtype)
    stop = np.full(cols, 0xFF, dtype=np.uint8)
    getattr(lib(), "oracle_gmres_initialize_" + _VT[b2.dtype])(
        _i64(rows), _i64(cols), _p(b2), _i64(cols), _p(res), _i64(cols), _p(gsin),
        _p(gcos), _i64(krylov_dim), _p(stop))
    return res, gsin, gcos, stop


def gmres_restart(residual, residual_norm, n_basis_rows):
    r2 = np.ascontiguousarray(_as2d(residual))
    rows, cols = r2.shape
    rn = np.ascontiguousarray(residual_norm, dtype=r2.dtype)
    rnc0 = np.zeros(cols, dtype=r2.dtype)
    krylov = np.full((n_basis_rows, cols), np.nan, dtype=r2.dtype)
    fin = np.full(cols, 99, dtype=np.uint64)
    getattr(lib(), "oracle_gmres_restart_" + _VT[r2.dtype])(
        _i64(rows), _i64(cols), _p(r2), _i64(cols), _p(rn), _p(rnc0), _p(krylov),
        _i64(cols), _p(fin))
    return rnc0, krylov, fin


def gmres_multi_axpy(krylov, y, rows, final_iter_nums, stop):
    k2 = np.ascontiguousarray(_as2d(krylov))
    y2 = np.ascontiguousarray(_as2d(y))
    cols = k2.shape[1]
    out = np.full((rows, cols), np.nan, dtype=k2.dtype)
    st = np.array(stop, dtype=np.uint8, copy=True)
    getattr(lib(), "oracle_gmres_multi_axpy_" + _VT[k2.dtype])(
        _i64(rows), _i64(cols), _p(k2), _i64(cols), _p(y2), _i64(cols), _p(out),
        _i64(cols), _p(np.ascontiguousarray(final_iter_nums, dtype=np.uint64)), _p(st))
    return out, st


def gmres_multi_dot(krylov, next_krylov, num_dots):
    k2 = np.ascontiguousarray(_as2d(krylov))
    n2 = np.ascontiguousarray(_as2d(next_krylov))
    rows, cols = n2.shape
    h = np.zeros((num_dots, cols), dtype=k2.dtype)
    getattr(lib(), "oracle_gmres_multi_dot_" + _VT[k2.dtype])(
        _i64(rows), _i64(cols), _i64(num_dots), _p(k2), _i64(cols), _p(n2), _i64(cols),
        _p(h), _i64(cols))
    return h


def gmres_hessenberg_qr(gsin, gcos, residual_norm, rnc, h, it, final_iter_nums, stop):
    arrs = [np.array(a, order="C", copy=True) for a in (gsin, gcos, residual_norm, rnc, h)]
    fin = np.array(final_iter_nums, dtype=np.uint64, copy=True)
    cols = arrs[0].shape[1]
    getattr(lib(), "oracle_gmres_hessenberg_qr_" + _VT[arrs[0].dtype])(
        _i64(cols), _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), _p(arrs[4]),
        _i64(it), _p(fin), _p(np.ascontiguousarray(stop, dtype=np.uint8)))
    return (*arrs, fin)


def gmres_solve_krylov(rnc, hessenberg, final_iter_nums, stop):
    r2 = np.ascontiguousarray(rnc)
    h2 = np.ascontiguousarray(hessenberg)
    cols = r2.shape[1]
    y = np.full((h2.shape[0], cols), np.nan, dtype=r2.dtype)
    getattr(lib(), "oracle_gmres_solve_krylov_" + _VT[r2.dtype])(
        _i64(cols), _p(r2), _p(h2), _i64(h2.shape[1]), _p(y),
        _p(np.ascontiguousarray(final_iter_nums, dtype=np.uint64)),
        _p(np.ascontiguousarray(stop, dtype=np.uint8)))
    return y
