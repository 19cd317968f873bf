"""TEST INFRASTRUCTURE -- ctypes front-end of oracle/_ref/libgko_ref_shim.so,
i.e. the UNMODIFIED reference (Ginkgo 1.12.0 ReferenceExecutor / OmpExecutor)
callable from the tests, the golden-fixture generator and bench.py's
cpu_baseline.  available() is False when oracle/_ref has not been built."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libgko_ref_shim.so")
_LIB = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(_PATH)
        _LIB.ref_version.restype = C.c_char_p
        _LIB.ref_csr_create.restype = C.c_void_p
        for n in ("ref_to_ell", "ref_to_sellp", "ref_jacobi_generate",
                  "ref_cg_solve", "ref_gmres_solve", "ref_stencil_subdomain"):
            getattr(_LIB, n).restype = C.c_int64
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def version():
    return lib().ref_version().decode()


class CsrHandle:
    """A gko::matrix::Csr<double,int32> on a Reference/Omp executor."""

    def __init__(self, exec_kind, row_ptrs, cols, vals, n_cols=None,
                 strategy="classical"):
        self.n_rows = len(row_ptrs) - 1
        self.n_cols = self.n_rows if n_cols is None else n_cols
        self._keep = (np.ascontiguousarray(row_ptrs, np.int32),
                      np.ascontiguousarray(cols, np.int32),
                      np.ascontiguousarray(vals, np.float64))
        self.h = C.c_void_p(lib().ref_csr_create(
            exec_kind.encode(), C.c_int64(self.n_rows), C.c_int64(self.n_cols),
            C.c_int64(len(vals)), _p(self._keep[0]), _p(self._keep[1]),
            _p(self._keep[2]), strategy.encode()))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_destroy(self.h)
            self.h = None

    def spmv(self, b, alpha=None, beta=None, c=None):
        b2 = np.ascontiguousarray(b if np.ndim(b) == 2 else np.reshape(b, (-1, 1)))
        nrhs = b2.shape[1]
        if alpha is None:
            out = np.empty((self.n_rows, nrhs))
            lib().ref_csr_spmv(self.h, _p(b2), C.c_int64(nrhs), _p(out),
                               C.c_int64(nrhs), C.c_int64(nrhs))
        else:
            out = np.array(c if np.ndim(c) == 2 else np.reshape(c, (-1, 1)),
                           dtype=np.float64, order="C", copy=True)
            lib().ref_csr_advanced_spmv(self.h, C.c_double(alpha), _p(b2),
                                        C.c_int64(nrhs), C.c_double(beta),
                                        _p(out), C.c_int64(nrhs), C.c_int64(nrhs))
        return out if np.ndim(b) == 2 else out[:, 0]

    def to_ell(self):
        k, st = C.c_int64(0), C.c_int64(0)
        n = lib().ref_to_ell(self.h, C.byref(k), C.byref(st))
        cols, vals = np.empty(n, np.int32), np.empty(n, np.float64)
        lib().ref_ell_get(self.h, _p(cols), _p(vals))
        return k.value, st.value, cols, vals

    def ell_spmv(self, b):
        b2 = np.ascontiguousarray(np.reshape(b, (len(b), -1)))
        out = np.empty((self.n_rows, b2.shape[1]))
        lib().ref_ell_spmv(self.h, _p(b2), _p(out), C.c_int64(b2.shape[1]))
        return out if np.ndim(b) == 2 else out[:, 0]

    def to_sellp(self, slice_size=64, stride_factor=1):
        ns = C.c_int64(0)
        n = lib().ref_to_sellp(self.h, C.c_int64(slice_size),
                               C.c_int64(stride_factor), C.byref(ns))
        sets = np.empty(ns.value + 1, np.uint64)
        lens = np.empty(ns.value, np.uint64)
        cols, vals = np.empty(n, np.int32), np.empty(n, np.float64)
        lib().ref_sellp_get(self.h, _p(sets), _p(lens), _p(cols), _p(vals))
        return sets, lens, cols, vals

    def sellp_spmv(self, b):
        b2 = np.ascontiguousarray(np.reshape(b, (len(b), -1)))
        out = np.empty((self.n_rows, b2.shape[1]))
        lib().ref_sellp_spmv(self.h, _p(b2), _p(out), C.c_int64(b2.shape[1]))
        return out if np.ndim(b) == 2 else out[:, 0]

    def jacobi_generate(self, max_block_size):
        scheme = np.zeros(3, np.int64)
        storage = C.c_int64(0)
        nb = lib().ref_jacobi_generate(self.h, C.c_uint32(max_block_size),
                                       _p(scheme), C.byref(storage))
        ptrs = np.empty(nb + 1, np.int32)
        blocks = np.empty(storage.value, np.float64)
        lib().ref_jacobi_get(self.h, _p(ptrs), _p(blocks))
        return nb, tuple(int(s) for s in scheme), ptrs, blocks

    def jacobi_generate_prec(self, max_block_size, preserving, nonpreserving):
        """Jacobi with storage_optimization = precision_reduction(preserving, nonpreserving);
        blocks = the raw storage (as float64 words) the reference holds"""
        scheme = np.zeros(3, np.int64)
        storage = C.c_int64(0)
        nb = lib().ref_jacobi_generate_prec(self.h, C.c_uint32(max_block_size), C.c_int(preserving),
                                            C.c_int(nonpreserving), _p(scheme), C.byref(storage))
        ptrs = np.empty(nb + 1, np.int32)
        blocks = np.empty(storage.value, np.float64)
        lib().ref_jacobi_get(self.h, _p(ptrs), _p(blocks))
        return nb, tuple(int(s) for s in scheme), ptrs, blocks

    def jacobi_generate_adaptive(self, max_block_size, accuracy=1e-1, requested=None):
        """adaptive Jacobi (autodetect, or the given per-block requests replicated over the
        blocks): (num_blocks, scheme, block_ptrs, raw blocks, chosen precisions, conditioning)"""
        scheme = np.zeros(3, np.int64)
        storage = C.c_int64(0)
        req = None if requested is None else np.ascontiguousarray(requested, np.uint8)
        f = lib().ref_jacobi_generate_adaptive
        f.restype = C.c_int64
        nb = f(self.h, C.c_uint32(max_block_size), C.c_double(accuracy), _p(req),
               C.c_int64(0 if req is None else len(req)), _p(scheme), C.byref(storage))
        ptrs = np.empty(nb + 1, np.int32)
        blocks = np.empty(storage.value, np.float64)
        lib().ref_jacobi_get(self.h, _p(ptrs), _p(blocks))
        prec, cond = np.zeros(nb, np.uint8), np.zeros(nb)
        lib().ref_jacobi_get_adaptive(self.h, _p(prec), _p(cond))
        return nb, tuple(int(s) for s in scheme), ptrs, blocks, prec, cond

    def jacobi_apply(self, b, alpha=None, beta=None, x=None):
        b2 = np.ascontiguousarray(np.reshape(b, (len(b), -1)))
        if alpha is None:
            out = np.empty_like(b2)
            lib().ref_jacobi_apply(self.h, _p(b2), _p(out), C.c_int64(b2.shape[1]),
                                   C.c_int(0), C.c_double(1), C.c_double(0))
        else:
            out = np.array(np.reshape(x, (len(x), -1)), dtype=np.float64,
                           order="C", copy=True)
            lib().ref_jacobi_apply(self.h, _p(b2), _p(out), C.c_int64(b2.shape[1]),
                                   C.c_int(1), C.c_double(alpha), C.c_double(beta))
        return out if np.ndim(b) == 2 else out[:, 0]

    def cg_solve(self, b, x0=None, max_iters=1000, reduction=1e-10,
                 baseline="rhs_norm", precond_block_size=0):
        x = np.zeros(self.n_rows) if x0 is None else np.array(x0, dtype=np.float64)
        b = np.ascontiguousarray(b, np.float64)
        rn = C.c_double(0)
        base = {"rhs_norm": 0, "initial_resnorm": 1, "absolute": 2}[baseline]
        it = lib().ref_cg_solve(self.h, C.c_uint32(precond_block_size), _p(b),
                                _p(x), C.c_int64(max_iters), C.c_double(reduction),
                                C.c_int(base), C.byref(rn))
        return x, int(it), rn.value

    def transpose(self):
        trp = np.zeros(self.n_cols + 1, np.int32)
        nnz = len(self._keep[2])
        tc, tv = np.zeros(nnz, np.int32), np.zeros(nnz)
        lib().ref_csr_transpose(self.h, _p(trp), _p(tc), _p(tv))
        return trp, tc, tv

    def to_hybrid(self, ell_lim):
        """Csr -> Hybrid(column_limit(ell_lim)); (ell_k, ell_stride, ell_cols, ell_vals,
        coo_rows, coo_cols, coo_vals)"""
        k, st = C.c_int64(0), C.c_int64(0)
        f = lib().ref_to_hybrid
        f.restype = C.c_int64
        nc = f(self.h, C.c_int64(ell_lim), C.byref(k), C.byref(st))
        ne = k.value * st.value
        ec, ev = np.zeros(ne, np.int32), np.zeros(ne)
        cr, cc, cv = np.zeros(nc, np.int32), np.zeros(nc, np.int32), np.zeros(nc)
        lib().ref_hybrid_get(self.h, _p(ec), _p(ev), _p(cr), _p(cc), _p(cv))
        return k.value, st.value, ec, ev, cr, cc, cv

    def hybrid_spmv(self, b):
        b2 = np.ascontiguousarray(b.reshape(len(b), -1), np.float64)
        out = np.zeros((self.n_rows, b2.shape[1]))
        lib().ref_hybrid_spmv(self.h, _p(b2), _p(out), C.c_int64(b2.shape[1]))
        return out if np.ndim(b) == 2 else out[:, 0]

    def gcr_solve(self, b, x0=None, krylov_dim=100, max_iters=1000, reduction=1e-10, precond_block_size=0):
        x = np.zeros(self.n_rows) if x0 is None else np.array(x0, dtype=np.float64)
        b = np.ascontiguousarray(b, np.float64)
        rn = C.c_double(0)
        f = lib().ref_gcr_solve
        f.restype = C.c_int64
        it = f(self.h, C.c_uint32(precond_block_size), _p(b), _p(x), C.c_int64(krylov_dim),
               C.c_int64(max_iters), C.c_double(reduction), C.byref(rn))
        return x, int(it), rn.value

    def stationary_solve(self, kind, b, x0=None, max_iters=1000, reduction=1e-10,
                         baseline="rhs_norm", precond_block_size=0, relaxation=1.0, foci=(0.0, 1.0)):
        """Ir (inner solver Jacobi / Identity) and Chebyshev of the reference"""
        x = np.zeros(self.n_rows) if x0 is None else np.array(x0, dtype=np.float64)
        b = np.ascontiguousarray(b, np.float64)
        rn = C.c_double(0)
        base = {"rhs_norm": 0, "initial_resnorm": 1, "absolute": 2}[baseline]
        f = lib().ref_stationary_solve
        f.restype = C.c_int64
        k = {"ir": 5, "chebyshev": 6}[kind]
        it = f(self.h, C.c_int(k), C.c_uint32(precond_block_size), _p(b), _p(x), C.c_int64(max_iters),
               C.c_double(reduction), C.c_int(base), C.c_double(relaxation if kind == "ir" else foci[0]),
               C.c_double(foci[1]), C.byref(rn))
        return x, int(it), rn.value

    KINDS = {"bicgstab": 1, "cgs": 2, "fcg": 3, "pipe_cg": 4, "bicg": 7, "minres": 9}

    def krylov_solve(self, kind, b, x0=None, max_iters=1000, reduction=1e-10,
                     baseline="rhs_norm", precond_block_size=0):
        """Bicgstab / Cgs / Fcg / PipeCg of the reference"""
        x = np.zeros(self.n_rows) if x0 is None else np.array(x0, dtype=np.float64)
        b = np.ascontiguousarray(b, np.float64)
        rn = C.c_double(0)
        base = {"rhs_norm": 0, "initial_resnorm": 1, "absolute": 2}[baseline]
        f = lib().ref_krylov_solve
        f.restype = C.c_int64
        it = f(self.h, C.c_int(self.KINDS[kind]), C.c_uint32(precond_block_size), _p(b), _p(x),
               C.c_int64(max_iters), C.c_double(reduction), C.c_int(base), C.byref(rn))
        return x, int(it), rn.value


def _gmres(self, b, x0=None, krylov_dim=100, ortho="mgs", max_iters=1000,
           reduction=1e-10, precond_block_size=0):
    x = np.zeros(self.n_rows) if x0 is None else np.array(x0, dtype=np.float64)
    b = np.ascontiguousarray(b, np.float64)
    rn = C.c_double(0)
    om = {"mgs": 0, "cgs": 1, "cgs2": 2}[ortho]
    it = lib().ref_gmres_solve(self.h, C.c_uint32(precond_block_size), _p(b), _p(x),
                               C.c_int64(krylov_dim), C.c_int(om), C.c_int64(max_iters),
                               C.c_double(reduction), C.byref(rn))
    return x, int(it), rn.value


CsrHandle.gmres_solve = _gmres


def dense_dot(x, y, exec_kind="reference"):
    x2 = np.ascontiguousarray(np.reshape(x, (len(x), -1)))
    y2 = np.ascontiguousarray(np.reshape(y, (len(y), -1)))
    res = np.zeros(x2.shape[1])
    lib().ref_dense_dot(exec_kind.encode(), C.c_int64(x2.shape[0]),
                        C.c_int64(x2.shape[1]), _p(x2), _p(y2), _p(res))
    return res


def dense_norm2(x, exec_kind="reference"):
    x2 = np.ascontiguousarray(np.reshape(x, (len(x), -1)))
    res = np.zeros(x2.shape[1])
    lib().ref_dense_norm2(exec_kind.encode(), C.c_int64(x2.shape[0]),
                          C.c_int64(x2.shape[1]), _p(x2), _p(res))
    return res


def stencil_subdomain(nd, dims, pos, target_local_size, restricted):
    """the reference's generate_{2,3}d_stencil_subdomain (COO, global idx)"""
    dims = np.asarray(dims, np.int32)
    pos = np.asarray(pos, np.int32)
    ls = C.c_int64(0)
    f = lib().ref_stencil_subdomain
    nnz = f(C.c_int(nd), _p(dims), _p(pos), C.c_int64(target_local_size),
            C.c_int(int(restricted)), None, None, None, C.byref(ls))
    rows, cols = np.empty(nnz, np.int64), np.empty(nnz, np.int64)
    vals = np.empty(nnz, np.float64)
    f(C.c_int(nd), _p(dims), _p(pos), C.c_int64(target_local_size),
      C.c_int(int(restricted)), _p(rows), _p(cols), _p(vals), C.byref(ls))
    return rows, cols, vals, int(ls.value)


def md_assemble(op, n_rows, n_cols, rows, cols, vals):
    """device_matrix_data::{sort_row_major, remove_zeros, sum_duplicates} of the reference, or
    op = "csr": Csr::read(device_matrix_data) after sort_row_major -> (row_ptrs, cols, vals)"""
    code = {"sort_row_major": 0, "remove_zeros": 1, "sum_duplicates": 2, "csr": 3}[op]
    nnz = len(vals)
    r = np.zeros(max(nnz, n_rows + 1), np.int32)
    r[:nnz] = rows
    c = np.array(cols, np.int32, copy=True)
    v = np.array(vals, np.float64, copy=True)
    f = lib().ref_md_assemble
    f.restype = C.c_int64
    n = f(C.c_int(code), C.c_int64(n_rows), C.c_int64(n_cols), C.c_int64(nnz), _p(r), _p(c), _p(v))
    if code == 3:
        return r[:n_rows + 1].copy(), c, v
    return r[:n].copy(), c[:n].copy(), v[:n].copy()


def coo_apply(mode, n_rows, n_cols, rows, cols, vals, b, alpha=1.0, beta=0.0, c=None, exec_kind="reference"):
    """Coo::apply / apply2 of the reference on the given index arrays"""
    m = {"spmv": 0, "advanced_spmv": 1, "spmv2": 2, "advanced_spmv2": 3}[mode]
    b2 = np.ascontiguousarray(np.asarray(b, np.float64).reshape(n_cols, -1))
    nrhs = b2.shape[1]
    out = np.zeros((n_rows, nrhs)) if c is None else \
        np.array(np.asarray(c, np.float64).reshape(n_rows, -1), order="C", copy=True)
    lib().ref_coo_apply(exec_kind.encode(), C.c_int(m), C.c_int64(n_rows), C.c_int64(n_cols),
                        C.c_int64(len(vals)), _p(np.ascontiguousarray(rows, np.int32)),
                        _p(np.ascontiguousarray(cols, np.int32)),
                        _p(np.ascontiguousarray(vals, np.float64)), C.c_double(alpha),
                        C.c_double(beta), _p(b2), _p(out), C.c_int64(nrhs))
    return out if np.ndim(b) == 2 else out[:, 0]
