"""Matrix formats on the Cdna4Executor: Dense, Csr, Ell, Sellp.

Host-side mirror of include/ginkgo/core/matrix/{dense,csr,ell,sellp}.hpp for
the hot path: same method names, argument meaning and error behaviour; every
numerical method is one call into libgko_cdna4.so (no torch arithmetic).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import DimensionMismatch, GkoError, IT, VT, call, cval
from .base import LinOp
from .executor import MEM_INDICES, MEM_VALUES, MEM_VECTOR


def _np_dtype(t):
    return {torch.float64: np.float64, torch.float32: np.float32,
            torch.int32: np.int32, torch.int64: np.int64,
            torch.uint8: np.uint8}[t]


class Dense(LinOp):
    """Row-major dense matrix with row stride (dense.hpp:88).  `values` is a
    2-D torch view whose row stride is the Ginkgo stride."""

    def __init__(self, exec_, values):
        assert values.dim() == 2
        if values.shape[1] > 0 and values.shape[0] > 0 and values.stride(1) != 1:
            raise GkoError("Dense: column stride must be 1")
        super().__init__(exec_, values.shape)
        self.values = values
        self._tmp = None

    # -- construction
    @staticmethod
    def create(exec_, size, dtype=torch.float64, stride=None, role=MEM_VECTOR):
        rows, cols = size
        stride = cols if stride is None else stride
        if stride < cols:
            raise GkoError("Dense: stride smaller than number of columns")
        store = exec_.alloc((rows, max(stride, 1)), dtype, role)
        return Dense(exec_, store[:, :cols])

    @staticmethod
    def from_numpy(exec_, array, stride=None):
        a = np.asarray(array)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        d = Dense.create(exec_, a.shape, torch.from_numpy(a[:0]).dtype, stride)
        d.values.copy_(torch.from_numpy(np.ascontiguousarray(a)))
        return d

    def to_numpy(self):
        return self.values.detach().cpu().numpy().copy()

    def clone(self):
        out = Dense.create(self.exec, self.size, self.dtype, self.stride)
        out.copy_from(self)
        return out

    def create_submatrix(self, rows, cols):
        """dense.hpp create_submatrix(span rows, span cols): a strided view."""
        return Dense(self.exec, self.values[rows[0]:rows[1], cols[0]:cols[1]])

    # -- accessors
    @property
    def dtype(self):
        return self.values.dtype

    @property
    def ld(self):
        s = self.values.stride(0)
        return s if s >= self.size[1] else max(self.size[1], 1)

    stride = ld

    def _suf(self):
        return VT[self.dtype]

    def _work(self, cols):
        need = _lib.lib().gkoc_reduction_workspace_bytes(
            C.c_int64(self.size[0]), C.c_int64(cols),
            C.c_size_t(self.values.element_size()))
        if self._tmp is None or self._tmp.numel() < need:
            self._tmp = self.exec.alloc((need,), torch.uint8)
        return self._tmp

    def _check_same(self, other):
        if other.size != self.size:
            raise DimensionMismatch(f"expected {self.size}, got {other.size}")

    def _check_alpha(self, alpha):
        if alpha.size[0] != 1 or alpha.size[1] not in (1, self.size[1]):
            raise DimensionMismatch(
                f"scalar must be 1 x 1 or 1 x {self.size[1]}, got {alpha.size}")

    # -- BLAS-1 (dense.hpp scale / add_scaled / compute_dot / compute_norm2)
    def fill(self, value):
        call("gkoc_dense_fill_" + self._suf(), self.exec.stream, self.size[0],
             self.size[1], self.values, self.ld, cval(self.dtype, value))
        return self

    def copy_from(self, other):
        self._check_same(other)
        call("gkoc_dense_copy_" + self._suf(), self.exec.stream, self.size[0],
             self.size[1], other.values, other.ld, self.values, self.ld)
        return self

    def scale(self, alpha):
        self._check_alpha(alpha)
        call("gkoc_dense_scale_" + self._suf(), self.exec.stream, self.size[0],
             self.size[1], alpha.values, alpha.size[1], self.values, self.ld)
        return self

    def inv_scale(self, alpha):
        self._check_alpha(alpha)
        call("gkoc_dense_inv_scale_" + self._suf(), self.exec.stream,
             self.size[0], self.size[1], alpha.values, alpha.size[1],
             self.values, self.ld)
        return self

    def add_scaled(self, alpha, b):
        self._check_alpha(alpha)
        self._check_same(b)
        call("gkoc_dense_add_scaled_" + self._suf(), self.exec.stream,
             self.size[0], self.size[1], alpha.values, alpha.size[1], b.values,
             b.ld, self.values, self.ld)
        return self

    def sub_scaled(self, alpha, b):
        self._check_alpha(alpha)
        self._check_same(b)
        call("gkoc_dense_sub_scaled_" + self._suf(), self.exec.stream,
             self.size[0], self.size[1], alpha.values, alpha.size[1], b.values,
             b.ld, self.values, self.ld)
        return self

    def compute_dot(self, b, result):
        self._check_same(b)
        if result.size != (1, self.size[1]):
            raise DimensionMismatch(f"result must be 1 x {self.size[1]}")
        w = self._work(self.size[1])
        call("gkoc_dense_compute_dot_" + self._suf(), self.exec.stream,
             self.size[0], self.size[1], self.values, self.ld, b.values, b.ld,
             result.values, w, C.c_size_t(w.numel()))
        return result

    compute_conj_dot = compute_dot  # real value types

    def compute_norm2(self, result):
        if result.size != (1, self.size[1]):
            raise DimensionMismatch(f"result must be 1 x {self.size[1]}")
        w = self._work(self.size[1])
        call("gkoc_dense_compute_norm2_" + self._suf(), self.exec.stream,
             self.size[0], self.size[1], self.values, self.ld, result.values, w,
             C.c_size_t(w.numel()))
        return result

    def compute_squared_norm2(self, result):
        if result.size != (1, self.size[1]):
            raise DimensionMismatch(f"result must be 1 x {self.size[1]}")
        w = self._work(self.size[1])
        call("gkoc_dense_compute_squared_norm2_" + self._suf(),
             self.exec.stream, self.size[0], self.size[1], self.values, self.ld,
             result.values, w, C.c_size_t(w.numel()))
        return result

    def row_gather(self, row_idxs, gathered):
        """dense.hpp row_gather: gathered(i, :) = self(row_idxs[i], :)."""
        if gathered.size != (row_idxs.numel(), self.size[1]):
            raise DimensionMismatch("row_gather: bad target size")
        if gathered.dtype != self.dtype:
            # dense::row_gather<ValueType, OutputType, IndexType> between two precisions
            # (core/matrix/dense_kernels.hpp:284-288)
            code = _SparseBase._CODE
            if self.dtype not in code or gathered.dtype not in code:
                raise _lib.NotSupported("row_gather: unsupported value types")
            call("gkoc_dense_row_gather_mixed_" + IT[row_idxs.dtype], self.exec.stream, C.c_int(code[self.dtype]),
                 C.c_int(code[gathered.dtype]), row_idxs.numel(), self.size[1], None, row_idxs, self.values,
                 self.ld, None, gathered.values, gathered.ld)
            return gathered
        call(f"gkoc_dense_row_gather_{self._suf()}_{IT[row_idxs.dtype]}",
             self.exec.stream, row_idxs.numel(), self.size[1], row_idxs,
             self.values, self.ld, gathered.values, gathered.ld)
        return gathered


def scalar(exec_, value, dtype=torch.float64):
    """1 x 1 device-resident Dense, like gko::initialize<Dense>({v}, exec)."""
    return Dense.from_numpy(exec_, np.array([[value]], dtype=_np_dtype(dtype)))


class _SparseBase(LinOp):
    def _operands(self, b, x):
        if b.dtype != self.dtype or x.dtype != self.dtype:
            raise _lib.NotSupported("mixed-precision apply: Csr and Ell only (as in the reference: "
                                    "core/matrix/{csr,ell}_kernels.hpp declare the triples)")
        return b.values, b.ld, x.values, x.ld, b.size[1]

    _CODE = {torch.float64: 0, torch.float32: 1, torch.complex128: 2, torch.complex64: 3}   # GKOC_VT_*

    def _non_uniform(self, b, x):
        return b.dtype != self.dtype or x.dtype != self.dtype

    def _apply_triple(self, fmt, alpha, b, beta, x):
        """csr / ell spmv for a non-uniform (matrix, input, output) value-type triple, as a core
        built with GINKGO_MIXED_PRECISION dispatches it (precision_dispatch.hpp:
        mixed_precision_dispatch_real_complex): alpha is taken in the matrix' type, beta in the
        output's (make_temporary_conversion), arithmetic in the widest type of the three"""
        codes = [self._CODE.get(t) for t in (self.dtype, b.dtype, x.dtype)]
        if None in codes or len({c >> 1 for c in codes}) != 1:
            raise _lib.NotSupported("mixed-precision apply: float32 / float64 or complex64 / complex128 triples")
        av = bv = None
        if alpha is not None:
            av = alpha.values if alpha.dtype == self.dtype else alpha.values.to(self.dtype)
            bv = beta.values if beta.dtype == x.dtype else beta.values.to(x.dtype)
        it = IT[self.col_idxs.dtype]
        codes = [C.c_int(c) for c in codes]
        if fmt == "csr":
            call("gkoc_csr_spmv_mixed_" + it, self.exec.stream, *codes, self.size[0], self.size[1], av,
                 self.row_ptrs, self.col_idxs, self.values, b.values, b.ld, bv, x.values, x.ld, b.size[1])
        else:
            call("gkoc_ell_spmv_mixed_" + it, self.exec.stream, *codes, self.size[0], self.size[1],
                 self.num_stored_per_row, self.stride, av, self.col_idxs, self.values, b.values, b.ld, bv,
                 x.values, x.ld, b.size[1])

    def _mixed(self, b, x, *scalars):
        """float32 values applied to float64 vectors (csr::spmv<float, double, double>, arithmetic
        in the highest precision): the kernels take the values as stored, 8 B per entry"""
        return (self.dtype == torch.float32 and b.dtype == torch.float64 and x.dtype == torch.float64
                and all(t.dtype == torch.float64 for t in scalars))


class Csr(_SparseBase):
    """csr.hpp:104.  The strategy argument is accepted for API compatibility
    (classical / load_balance / merge_path / sparselib / automatical): this
    backend has one row-segment-per-wavefront kernel for all of them."""

    def __init__(self, exec_, size, values, col_idxs, row_ptrs,
                 strategy="automatical"):
        super().__init__(exec_, size)
        if row_ptrs.numel() != self.size[0] + 1:
            raise GkoError("Csr: row_ptrs must have num_rows + 1 entries")
        if values.numel() != col_idxs.numel():
            raise GkoError("Csr: values / col_idxs length mismatch")
        if col_idxs.dtype != row_ptrs.dtype:
            raise GkoError("Csr: index arrays must share one type")
        self.values, self.col_idxs, self.row_ptrs = values, col_idxs, row_ptrs
        self.strategy = strategy

    @staticmethod
    def from_scipy(exec_, a, index_dtype=np.int32, strategy="automatical"):
        a = a.tocsr()
        return Csr(exec_, a.shape, exec_.to_device(np.asarray(a.data), MEM_VALUES),
                   exec_.to_device(a.indices.astype(index_dtype), MEM_INDICES),
                   exec_.to_device(a.indptr.astype(index_dtype), MEM_INDICES), strategy)

    @staticmethod
    def read(data, strategy="automatical"):
        """Csr::read(const device_matrix_data&) (core/matrix/csr.cpp:583-606): the entries
        must be sorted row-major; values and column indices are taken as they are, the
        row pointers come from components::convert_idxs_to_ptrs on the device"""
        ex, n = data.exec, data.size[0]
        ptrs = ex.alloc((n + 1,), data.row_idxs.dtype, MEM_INDICES)
        call("gkoc_convert_idxs_to_ptrs_" + IT[data.row_idxs.dtype], ex.stream,
             data.get_num_stored_elements(), data.row_idxs, n, ptrs)
        return Csr(ex, data.size, data.values, data.col_idxs, ptrs, strategy)

    @staticmethod
    def from_arrays(exec_, size, row_ptrs, col_idxs, values,
                    strategy="automatical"):
        return Csr(exec_, size, exec_.to_device(values, MEM_VALUES),
                   exec_.to_device(col_idxs, MEM_INDICES),
                   exec_.to_device(row_ptrs, MEM_INDICES), strategy)

    @property
    def dtype(self):
        return self.values.dtype

    def _suf(self):
        return f"{VT[self.values.dtype]}_{IT[self.col_idxs.dtype]}"

    def get_num_stored_elements(self):
        return self.values.numel()

    def apply_impl(self, b, x):
        if self._mixed(b, x):
            call("gkoc_csr_spmv_f32_f64_" + IT[self.col_idxs.dtype], self.exec.stream, self.size[0],
                 self.size[1], self.row_ptrs, self.col_idxs, self.values, b.values, b.ld, x.values,
                 x.ld, b.size[1])
            return
        if self._non_uniform(b, x):
            return self._apply_triple("csr", None, b, None, x)
        bv, ldb, xv, ldx, nrhs = self._operands(b, x)
        call("gkoc_csr_spmv_" + self._suf(), self.exec.stream, self.size[0],
             self.size[1], self.row_ptrs, self.col_idxs, self.values, bv, ldb,
             xv, ldx, nrhs)

    def apply_advanced_impl(self, alpha, b, beta, x):
        if self._mixed(b, x, alpha, beta):
            call("gkoc_csr_advanced_spmv_f32_f64_" + IT[self.col_idxs.dtype], self.exec.stream,
                 self.size[0], self.size[1], alpha.values, self.row_ptrs, self.col_idxs, self.values,
                 b.values, b.ld, beta.values, x.values, x.ld, b.size[1])
            return
        if self._non_uniform(b, x):
            return self._apply_triple("csr", alpha, b, beta, x)
        bv, ldb, xv, ldx, nrhs = self._operands(b, x)
        call("gkoc_csr_advanced_spmv_" + self._suf(), self.exec.stream,
             self.size[0], self.size[1], alpha.values, self.row_ptrs,
             self.col_idxs, self.values, bv, ldb, beta.values, xv, ldx, nrhs)

    def apply_dot(self, b, x, dot_out, work):
        """x = A b and dot_out = <b, x> in one pass (gkoc_x_csr_spmv_dot_*):
        square matrix, one right-hand side, unit strides"""
        call("gkoc_x_csr_spmv_dot_" + self._suf(), self.exec.stream, self.size[0],
             self.row_ptrs, self.col_idxs, self.values, b.values, x.values,
             dot_out.values, work, C.c_size_t(work.numel() * work.element_size()))

    def extract_diagonal(self):
        n = min(self.size)
        d = self.exec.zeros((n,), self.dtype)
        call("gkoc_csr_extract_diagonal_" + self._suf(), self.exec.stream,
             self.size[0], self.size[1], self.row_ptrs, self.col_idxs,
             self.values, d)
        return d

    def is_sorted_by_column_index(self):
        flag = C.c_int(1)
        call("gkoc_csr_is_sorted_by_column_index_" + self._suf(),
             self.exec.stream, self.size[0], self.row_ptrs, self.col_idxs,
             C.byref(flag))
        return bool(flag.value)

    def sort_by_column_index(self):
        call("gkoc_csr_sort_by_column_index_" + self._suf(), self.exec.stream,
             self.size[0], self.row_ptrs, self.col_idxs, self.values)
        return self

    def convert_to_ell(self, num_stored_per_row=None, stride=None):
        it = IT[self.col_idxs.dtype]
        if num_stored_per_row is None:
            m = C.c_int64(0)
            call("gkoc_compute_max_row_nnz_" + it, self.exec.stream,
                 self.size[0], self.row_ptrs, C.byref(m))
            num_stored_per_row = m.value
        stride = self.size[0] if stride is None else stride
        cols = self.exec.alloc((num_stored_per_row * stride,), self.col_idxs.dtype, MEM_INDICES)
        vals = self.exec.alloc((num_stored_per_row * stride,), self.dtype, MEM_VALUES)
        if stride > self.size[0]:
            cols.fill_(-1)
            vals.zero_()
        call("gkoc_csr_convert_to_ell_" + self._suf(), self.exec.stream,
             self.size[0], self.row_ptrs, self.col_idxs, self.values,
             num_stored_per_row, stride, cols, vals)
        return Ell(self.exec, self.size, vals, cols, num_stored_per_row, stride)

    def _time_apply(self, x, y, reps):
        self.apply(x, y)
        torch.cuda.synchronize(self.exec.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            self.apply(x, y)
        e1.record()
        torch.cuda.synchronize(self.exec.device)
        return e0.elapsed_time(e1) / reps

    def memory_classes(self):
        """memory class (arena) of values / col_idxs / row_ptrs; the SpMV is
        fastest when the output vector's class differs from those of values and col_idxs"""
        ex = self.exec
        return {"values": ex.memory_class(self.values), "col_idxs": ex.memory_class(self.col_idxs),
                "row_ptrs": ex.memory_class(self.row_ptrs)}

    def transpose(self):
        """Csr::transpose (core/matrix/csr.cpp, csr::transpose kernel)"""
        ex = self.exec
        nnz = int(self.col_idxs.numel())
        idt = self.col_idxs.dtype
        t_ptrs = ex.alloc((self.size[1] + 1,), idt, MEM_INDICES)
        t_cols, t_vals = ex.alloc((nnz,), idt, MEM_INDICES), ex.alloc((nnz,), self.dtype, MEM_VALUES)
        f = _lib.lib().gkoc_csr_transpose_workspace_bytes
        f.restype = C.c_size_t
        need = f(C.c_int64(nnz), C.c_int64(self.size[1]), C.c_size_t(self.col_idxs.element_size()))
        work = ex.alloc((int(need),), torch.uint8)
        call("gkoc_csr_transpose_" + self._suf(), ex.stream, self.size[0], self.size[1],
             self.row_ptrs, self.col_idxs, self.values, nnz, t_ptrs, t_cols, t_vals, work,
             C.c_size_t(work.numel()))
        ex.synchronize()        # `work` is released on return
        return Csr(ex, (self.size[1], self.size[0]), t_vals, t_cols, t_ptrs)

    def convert_to_coo(self):
        rows = self.exec.alloc((self.col_idxs.numel(),), self.col_idxs.dtype, MEM_INDICES)
        call("gkoc_convert_ptrs_to_idxs_" + IT[self.col_idxs.dtype], self.exec.stream,
             self.row_ptrs, self.size[0], rows)
        return Coo(self.exec, self.size, self.values, self.col_idxs, rows)

    def convert_to_hybrid(self, column_limit=None, imbalance_percent=0.8):
        """Csr::convert_to(Hybrid) (core/matrix/csr.cpp:417-441).  The split is the
        strategy's: column_limit(k) keeps k entries per row in the Ell part;
        otherwise imbalance_limit(percent) = the row length at that quantile of the
        sorted row lengths (hybrid.hpp:231-252; decided on the host like there)."""
        ex, n = self.exec, self.size[0]
        it = IT[self.col_idxs.dtype]
        sizes = ex.alloc((max(n, 1),), torch.int64)[:n]      # uint64 bits
        call("gkoc_convert_ptrs_to_sizes_" + it, ex.stream, n, self.row_ptrs, sizes)
        if column_limit is None:
            if n == 0:
                ell_lim = 0
            else:
                srt = torch.sort(sizes).values
                p = min(max(float(imbalance_percent), 0.0), 1.0)
                ell_lim = int(srt[int(n * p)].item()) if p < 1 else int(srt[-1].item())
        else:
            ell_lim = int(column_limit)
        ell_lim = min(ell_lim, self.size[1])
        crp = ex.alloc((n + 1,), torch.int64)
        call("gkoc_hybrid_compute_coo_row_ptrs", ex.stream, n, sizes, C.c_uint64(ell_lim), crp)
        coo_nnz = int(crp[-1].item())
        idt, vdt = self.col_idxs.dtype, self.dtype
        ec, ev = ex.alloc((ell_lim * n,), idt, MEM_INDICES), ex.alloc((ell_lim * n,), vdt, MEM_VALUES)
        cr, cc = ex.alloc((coo_nnz,), idt, MEM_INDICES), ex.alloc((coo_nnz,), idt, MEM_INDICES)
        cv = ex.alloc((coo_nnz,), vdt, MEM_VALUES)
        call("gkoc_csr_convert_to_hybrid_" + self._suf(), ex.stream, n, self.row_ptrs,
             self.col_idxs, self.values, ell_lim, n, ec, ev, crp, cr, cc, cv)
        hyb = Hybrid(ex, self.size, Ell(ex, self.size, ev, ec, ell_lim, n),
                     Coo(ex, self.size, cv, cc, cr))
        hyb.coo_row_ptrs = crp
        return hyb

    def convert_to_sellp(self, slice_size=64, stride_factor=1):
        it = IT[self.col_idxs.dtype]
        n_slices = (self.size[0] + slice_size - 1) // slice_size
        # uint64 arrays are carried as int64 tensors (same bits)
        sets = self.exec.zeros((n_slices + 1,), torch.int64)
        lens = self.exec.zeros((max(n_slices, 1),), torch.int64)[:n_slices]
        call("gkoc_sellp_compute_slice_sets_" + it, self.exec.stream,
             self.size[0], slice_size, stride_factor, self.row_ptrs, sets, lens)
        total = int(sets[-1].item()) * slice_size
        cols = self.exec.alloc((total,), self.col_idxs.dtype, MEM_INDICES)
        vals = self.exec.alloc((total,), self.dtype, MEM_VALUES)
        if total:
            cols.fill_(-1)   # rows past num_rows in the last slice stay padding
            vals.zero_()
        call("gkoc_csr_convert_to_sellp_" + self._suf(), self.exec.stream,
             self.size[0], slice_size, self.row_ptrs, self.col_idxs,
             self.values, sets, cols, vals)
        return Sellp(self.exec, self.size, vals, cols, sets, lens, slice_size,
                     stride_factor)


class Coo(_SparseBase):
    """coo.hpp: (row, column, value) triplets, rows ascending.  apply = c = A b;
    apply2 = c += A b (Coo::apply2, coo.hpp) as Hybrid uses it."""

    def __init__(self, exec_, size, values, col_idxs, row_idxs):
        super().__init__(exec_, size)
        self.values, self.col_idxs, self.row_idxs = values, col_idxs, row_idxs
        need = _lib.lib().gkoc_coo_workspace_bytes(C.c_int64(size[0]),
                                                   C.c_size_t(col_idxs.element_size()),
                                                   C.c_size_t(values.element_size()))
        self._work = exec_.alloc((int(need),), torch.uint8)

    @staticmethod
    def read(data):
        """Coo::read(const device_matrix_data&) (core/matrix/coo.cpp): the arrays as they are"""
        return Coo(data.exec, data.size, data.values, data.col_idxs, data.row_idxs)

    @property
    def dtype(self):
        return self.values.dtype

    def _suf(self):
        return f"{VT[self.values.dtype]}_{IT[self.col_idxs.dtype]}"

    def get_num_stored_elements(self):
        return int(self.values.numel())

    def _run(self, name, alpha, b, beta, x):
        bv, ldb, xv, ldx, nrhs = self._operands(b, x)
        args = [self.exec.stream, self.size[0], self.size[1], self.values.numel()]
        if alpha is not None:
            args.append(alpha.values)
        args += [self.row_idxs, self.col_idxs, self.values, bv, ldb]
        if beta is not None:
            args.append(beta.values)
        args += [xv, ldx, nrhs, self._work, C.c_size_t(self._work.numel())]
        call(name + self._suf(), *args)

    def apply_impl(self, b, x):
        self._run("gkoc_coo_spmv_", None, b, None, x)

    def apply_advanced_impl(self, alpha, b, beta, x):
        self._run("gkoc_coo_advanced_spmv_", alpha, b, beta, x)

    def apply2(self, *args):
        """apply2(b, x): x += A b ; apply2(alpha, b, x): x += alpha A b"""
        if len(args) == 2:
            self._run("gkoc_coo_spmv2_", None, args[0], None, args[1])
        else:
            self._run("gkoc_coo_advanced_spmv2_", args[0], args[1], None, args[2])
        return args[-1]


def entry_dtype(value_dtype, index_dtype):
    """numpy layout of gko::matrix_data_entry<V, I> = struct { I row; I column; V value; }
    with natural alignment (include/ginkgo/core/base/matrix_data.hpp:60)"""
    return np.dtype([("row", index_dtype), ("column", index_dtype), ("value", value_dtype)],
                    align=True)


class DeviceMatrixData:
    """gko::device_matrix_data<V, I> (include/ginkgo/core/base/device_matrix_data.hpp:36;
    core/base/device_matrix_data.cpp): the structure-of-arrays triplets a matrix is
    assembled from ON THE DEVICE - sort_row_major, remove_zeros and sum_duplicates run
    there, and Csr.read(data) / Coo.read(data) take the arrays over without a host trip."""

    def __init__(self, exec_, size, row_idxs, col_idxs, values):
        if not (row_idxs.numel() == col_idxs.numel() == values.numel()):
            raise GkoError("device_matrix_data: arrays differ in length")
        if row_idxs.dtype != col_idxs.dtype:
            raise GkoError("device_matrix_data: index arrays must share one type")
        self.exec, self.size = exec_, (int(size[0]), int(size[1]))
        self.row_idxs, self.col_idxs, self.values = row_idxs, col_idxs, values

    @staticmethod
    def create_from_host(exec_, size, entries):
        """device_matrix_data::create_from_host (device_matrix_data.cpp:60-72): `entries`
        is a structured array of entry_dtype (the nonzeros of a gko::matrix_data); they
        are copied to the device as they are and split there (components::aos_to_soa)"""
        entries = np.ascontiguousarray(entries)
        vdt, idt = entries.dtype["value"], entries.dtype["row"]
        if entries.dtype != entry_dtype(vdt, idt):
            raise GkoError("create_from_host: entries are not laid out like matrix_data_entry")
        nnz = len(entries)
        raw = exec_.to_device(entries.view(np.uint8))
        tv = torch.from_numpy(np.empty(0, vdt)).dtype
        ti = torch.from_numpy(np.empty(0, idt)).dtype
        rows, cols, vals = exec_.alloc((nnz,), ti), exec_.alloc((nnz,), ti), exec_.alloc((nnz,), tv)
        call(f"gkoc_aos_to_soa_{VT[tv]}_{IT[ti]}", exec_.stream, nnz, raw, rows, cols, vals)
        exec_.synchronize()     # `raw` is released on return
        return DeviceMatrixData(exec_, size, rows, cols, vals)

    def _suf(self):
        return f"{VT[self.values.dtype]}_{IT[self.col_idxs.dtype]}"

    def get_num_stored_elements(self):
        return int(self.values.numel())

    def copy_to_host(self):
        """device_matrix_data::copy_to_host (components::soa_to_aos): the entries as a
        structured array of entry_dtype"""
        nnz = self.get_num_stored_elements()
        dt = entry_dtype(_np_dtype(self.values.dtype), _np_dtype(self.col_idxs.dtype))
        raw = self.exec.alloc((nnz * dt.itemsize,), torch.uint8)
        call("gkoc_soa_to_aos_" + self._suf(), self.exec.stream, nnz, self.row_idxs,
             self.col_idxs, self.values, raw)
        self.exec.synchronize()
        return raw.cpu().numpy().view(dt)

    def sort_row_major(self):
        """stable by (row, column), in place (device_matrix_data.cpp:106-112)"""
        nnz = self.get_num_stored_elements()
        f = _lib.lib().gkoc_sort_row_major_workspace_bytes
        f.restype = C.c_size_t
        need = f(C.c_int64(nnz), C.c_size_t(self.values.element_size()),
                 C.c_size_t(self.col_idxs.element_size()))
        work = self.exec.alloc((int(need),), torch.uint8)
        call("gkoc_sort_row_major_" + self._suf(), self.exec.stream, nnz, self.row_idxs,
             self.col_idxs, self.values, work, C.c_size_t(work.numel()))
        self.exec.synchronize()     # `work` is released on return
        return self

    def _compact(self, count_name, count_args, fill_name):
        nnz = self.get_num_stored_elements()
        f = _lib.lib().gkoc_compact_workspace_bytes
        f.restype = C.c_size_t
        work = self.exec.alloc((int(f(C.c_int64(nnz))),), torch.uint8)
        kept = C.c_int64(0)
        call(count_name, self.exec.stream, nnz, *count_args, work, C.c_size_t(work.numel()),
             C.byref(kept))
        if kept.value < nnz:        # otherwise the arrays stay as they are, like the reference's
            ex, n = self.exec, kept.value
            rows, cols = ex.alloc((n,), self.row_idxs.dtype), ex.alloc((n,), self.col_idxs.dtype)
            vals = ex.alloc((n,), self.values.dtype)
            call(fill_name + self._suf(), ex.stream, nnz, self.row_idxs, self.col_idxs,
                 self.values, work, rows, cols, vals)
            ex.synchronize()        # `work` and the old arrays are released
            self.row_idxs, self.col_idxs, self.values = rows, cols, vals
        return self

    def remove_zeros(self):
        """drops the entries whose value == 0 (device_matrix_data.cpp:115-120)"""
        return self._compact("gkoc_remove_zeros_count_" + VT[self.values.dtype], (self.values,),
                             "gkoc_remove_zeros_fill_")

    def sum_duplicates(self):
        """sort_row_major, then one entry per (row, column) whose value is the sum of the
        run in storage order (device_matrix_data.cpp:123-129)"""
        self.sort_row_major()
        return self._compact("gkoc_sum_duplicates_count_" + IT[self.col_idxs.dtype],
                             (self.row_idxs, self.col_idxs), "gkoc_sum_duplicates_fill_")


class Hybrid(_SparseBase):
    """hybrid.hpp: an Ell part holding the first ell_lim entries of every row and a
    Coo part with the rest; apply = ell.apply then coo.apply2 (core/matrix/hybrid.cpp)."""

    def __init__(self, exec_, size, ell, coo):
        super().__init__(exec_, size)
        self.ell, self.coo = ell, coo

    @property
    def dtype(self):
        return self.ell.dtype

    def get_num_stored_elements(self):
        return int(self.ell.values.numel()) + self.coo.get_num_stored_elements()

    def apply_impl(self, b, x):
        self.ell.apply(b, x)
        self.coo.apply2(b, x)

    def apply_advanced_impl(self, alpha, b, beta, x):
        self.ell.apply(alpha, b, beta, x)
        self.coo.apply2(alpha, b, x)


class Ell(_SparseBase):
    """ell.hpp: column-major, entry (row, j) at row + j*stride, padding -1."""

    def __init__(self, exec_, size, values, col_idxs, num_stored_per_row,
                 stride):
        super().__init__(exec_, size)
        self.values, self.col_idxs = values, col_idxs
        self.num_stored_per_row, self.stride = int(num_stored_per_row), int(stride)

    @property
    def dtype(self):
        return self.values.dtype

    def _suf(self):
        return f"{VT[self.values.dtype]}_{IT[self.col_idxs.dtype]}"

    def apply_impl(self, b, x):
        if self._mixed(b, x):
            call("gkoc_ell_spmv_f32_f64_" + IT[self.col_idxs.dtype], self.exec.stream, self.size[0],
                 self.size[1], self.num_stored_per_row, self.stride, self.col_idxs, self.values,
                 b.values, b.ld, x.values, x.ld, b.size[1])
            return
        if self._non_uniform(b, x):
            return self._apply_triple("ell", None, b, None, x)
        bv, ldb, xv, ldx, nrhs = self._operands(b, x)
        call("gkoc_ell_spmv_" + self._suf(), self.exec.stream, self.size[0],
             self.size[1], self.num_stored_per_row, self.stride, self.col_idxs,
             self.values, bv, ldb, xv, ldx, nrhs)

    def apply_advanced_impl(self, alpha, b, beta, x):
        if self._mixed(b, x, alpha, beta):
            call("gkoc_ell_advanced_spmv_f32_f64_" + IT[self.col_idxs.dtype], self.exec.stream,
                 self.size[0], self.size[1], self.num_stored_per_row, self.stride, alpha.values,
                 self.col_idxs, self.values, b.values, b.ld, beta.values, x.values, x.ld, b.size[1])
            return
        if self._non_uniform(b, x):
            return self._apply_triple("ell", alpha, b, beta, x)
        bv, ldb, xv, ldx, nrhs = self._operands(b, x)
        call("gkoc_ell_advanced_spmv_" + self._suf(), self.exec.stream,
             self.size[0], self.size[1], self.num_stored_per_row, self.stride,
             alpha.values, self.col_idxs, self.values, bv, ldb, beta.values, xv,
             ldx, nrhs)


class Sellp(_SparseBase):
    """sellp.hpp:27 (slice_size 64 = one wavefront per slice)."""

    def __init__(self, exec_, size, values, col_idxs, slice_sets,
                 slice_lengths, slice_size=64, stride_factor=1):
        super().__init__(exec_, size)
        self.values, self.col_idxs = values, col_idxs
        self.slice_sets, self.slice_lengths = slice_sets, slice_lengths
        self.slice_size, self.stride_factor = int(slice_size), int(stride_factor)

    @property
    def dtype(self):
        return self.values.dtype

    def _suf(self):
        return f"{VT[self.values.dtype]}_{IT[self.col_idxs.dtype]}"

    def apply_impl(self, b, x):
        bv, ldb, xv, ldx, nrhs = self._operands(b, x)
        call("gkoc_sellp_spmv_" + self._suf(), self.exec.stream, self.size[0],
             self.size[1], self.slice_size, self.slice_sets, self.slice_lengths,
             self.col_idxs, self.values, bv, ldb, xv, ldx, nrhs)

    def apply_advanced_impl(self, alpha, b, beta, x):
        bv, ldb, xv, ldx, nrhs = self._operands(b, x)
        call("gkoc_sellp_advanced_spmv_" + self._suf(), self.exec.stream,
             self.size[0], self.size[1], self.slice_size, alpha.values,
             self.slice_sets, self.slice_lengths, self.col_idxs, self.values,
             bv, ldb, beta.values, xv, ldx, nrhs)


def stencil_csr(exec_, nd, g, restricted=False, dtype=torch.float64,
                index_dtype=torch.int32, z0=0, nz=None):
    """Benchmark stencil matrix generated on the device
    (benchmark/utils/stencil_matrix.hpp:68-453).  Returns the Csr of the rows
    of planes [z0, z0+nz) with global column indices (default: whole matrix)."""
    nz = g if nz is None else nz
    n_local = nz * (g * g if nd == 3 else g)
    n_global = g ** nd
    it, vt = IT[index_dtype], VT[dtype]
    row_ptrs = exec_.alloc((n_local + 1,), index_dtype, MEM_INDICES)
    nnz = C.c_int64(0)
    call("gkoc_stencil_row_ptrs_" + it, exec_.stream, C.c_int(nd), g,
         C.c_int(int(restricted)), z0, nz, row_ptrs, C.byref(nnz))
    if index_dtype == torch.int32 and nnz.value >= 2 ** 31:
        raise GkoError("stencil_csr: nnz overflows int32")
    cols = exec_.alloc((nnz.value,), index_dtype, MEM_INDICES)
    vals = exec_.alloc((nnz.value,), dtype, MEM_VALUES)
    call(f"gkoc_stencil_fill_{vt}_{it}", exec_.stream, C.c_int(nd), g,
         C.c_int(int(restricted)), z0, nz, row_ptrs, cols, vals)
    return Csr(exec_, (n_local, n_global), vals, cols, row_ptrs)
