"""Jacobi preconditioner (scalar and block) - mirror of
include/ginkgo/core/preconditioner/jacobi.hpp and
core/preconditioner/jacobi.cpp:150-165 (apply) / :328-404 (generate).

`Jacobi.build().with_max_block_size(8).on(exec).generate(A)` detects the
blocks (jacobi::find_blocks), inverts them (jacobi::generate) and applies them
(jacobi::simple_apply / apply), all on the device through libgko_cdna4.so.
max_block_size == 1 takes Ginkgo's scalar path (extract_diagonal +
invert_diagonal + simple_scalar_apply).  Adaptive precision
(`with_storage_optimization`) is supported for fp64 and fp32 values: block-wise and
autodetected precisions for any max_block_size <= 32, one fixed reduced
precision for all blocks for max_block_size in {2, 4, 8, 16}.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import IT, VT, JacobiScheme, NotSupported, call
from .base import LinOp
from .executor import MEM_INDICES
from .matrix import Csr


def compute_storage_scheme(max_block_size, warp_size=64):
    """jacobi.hpp:589-627 with max_block_stride = warp size (64 on this GPU)."""
    if max_block_size < 1 or max_block_size > warp_size:
        raise NotSupported("max_block_size must be in [1, 64]")
    p2 = 1
    while p2 < max_block_size:
        p2 *= 2
    group_size = warp_size // p2
    block_offset = max_block_size
    group_offset = max_block_size * group_size * block_offset
    return JacobiScheme(block_offset, group_offset, group_size.bit_length() - 1)


class JacobiFactory:
    def __init__(self):
        self.max_block_size = 32
        self.skip_sorting = False
        self.block_pointers = None
        self.storage_precision = 0
        self.block_wise = None
        self.accuracy = 1e-1
        self.exec = None

    def with_max_block_size(self, v):
        self.max_block_size = int(v)
        return self

    def with_skip_sorting(self, v):
        self.skip_sorting = bool(v)
        return self

    def with_block_pointers(self, ptrs):
        self.block_pointers = ptrs
        return self

    def with_storage_optimization(self, preserving, nonpreserving=None):
        """Jacobi's storage_optimization (jacobi.hpp:389-484):
        with_storage_optimization(p, n)      precision_reduction(p, n) for every block -
            (0,1) float, (0,2) half, (1,0) / (2,0) upper 32 / 16 bits of the double,
            (1,1) upper 16 bits of the float, (0,0) full precision;
        with_storage_optimization("autodetect")   per storage group, the smallest type
            whose unit round-off times the block condition number stays below
            with_accuracy (default 0.1);
        with_storage_optimization([...])     block-wise requests: (p, n) pairs or
            "autodetect", replicated over the blocks."""
        def byte(v):
            if v in ("autodetect", "auto"):
                return 0xff
            p_, n_ = v
            code = (int(p_) << 4) | int(n_)
            if code not in (0x00, 0x01, 0x02, 0x10, 0x11, 0x20):
                raise NotSupported(f"block-Jacobi: no storage type for precision_reduction({p_}, {n_})")
            return code
        self.storage_precision, self.block_wise = 0, None
        if nonpreserving is not None:
            self.storage_precision = byte((preserving, nonpreserving))
        elif preserving in ("autodetect", "auto"):
            self.block_wise = [0xff]
        elif isinstance(preserving, (list, tuple)) and preserving and \
                isinstance(preserving[0], (list, tuple, str)):
            self.block_wise = [byte(v) for v in preserving]
        else:
            self.storage_precision = byte(preserving)
        return self

    def with_accuracy(self, v):
        self.accuracy = float(v)
        return self

    def on(self, exec_):
        self.exec = exec_
        return self

    def generate(self, system_matrix):
        return Jacobi(self, system_matrix)


class Jacobi(LinOp):
    @staticmethod
    def build():
        return JacobiFactory()

    def __init__(self, factory, a):
        if not isinstance(a, Csr):
            raise NotSupported("Jacobi.generate needs a Csr system matrix")
        if a.size[0] != a.size[1]:
            from ._lib import DimensionMismatch
            raise DimensionMismatch("Jacobi needs a square matrix")
        super().__init__(factory.exec or a.exec, a.size)
        ex = self.exec
        self.max_block_size = factory.max_block_size
        self.dtype = a.dtype
        self.storage_precision = factory.storage_precision
        self.precisions = self.conditioning = None
        block_wise = factory.block_wise
        adaptive = block_wise is not None
        if a.dtype != torch.float64 and self.storage_precision:
            # one reduced precision for every block of a float matrix: that request block by block
            # (Jacobi::generate replicates it the same way, core/preconditioner/jacobi.cpp:386-396) -
            # the generic kernels of csrc/jacobi.hip; the in-place conversion below is fp64's
            block_wise, adaptive, self.storage_precision = [self.storage_precision], True, 0
        if self.storage_precision and self.max_block_size not in (2, 4, 8, 16):
            # the in-place conversion of ONE precision for all blocks works on 64-wide groups;
            # block-wise / autodetected precisions take any max_block_size <= 32
            raise NotSupported("block-Jacobi: one reduced storage precision for all blocks needs "
                               "max_block_size in {2, 4, 8, 16} (64-wide storage groups)")
        if adaptive and not 2 <= self.max_block_size <= 32:
            raise NotSupported("block-Jacobi: adaptive precision needs 2 <= max_block_size <= 32")
        self._suf = f"{VT[a.dtype]}_{IT[a.col_idxs.dtype]}"
        n = a.size[0]
        if not factory.skip_sorting and not a.is_sorted_by_column_index():
            a = Csr(ex, a.size, a.values.clone(), a.col_idxs.clone(),
                    a.row_ptrs, a.strategy).sort_by_column_index()
        if self.max_block_size == 1:
            # scalar Jacobi (jacobi.cpp:340-352)
            diag = a.extract_diagonal()
            self.inv_diag = ex.alloc((n,), a.dtype)
            call("gkoc_jacobi_invert_diagonal_" + VT[a.dtype], ex.stream, n,
                 diag, self.inv_diag)
            self.num_blocks = n
            return
        self.scheme = compute_storage_scheme(self.max_block_size,
                                             ex.get_warp_size())
        if factory.block_pointers is not None:
            # user-supplied blocks (jacobi.hpp:377-387): the kernels are selected by the
            # matrix' index type, so the array is converted to it; checked like
            # jacobi.cpp:358-363 expects them (ascending, covering all rows, <= max size)
            if isinstance(factory.block_pointers, torch.Tensor):
                hp = factory.block_pointers.detach().cpu().numpy()
            else:
                hp = np.asarray(factory.block_pointers)
            hp = hp.astype(np.int64).reshape(-1)
            sizes = np.diff(hp)
            if (hp.size < 1 or hp[0] != 0 or hp[-1] != n or (sizes < 0).any()
                    or (sizes > self.max_block_size).any()):
                from ._lib import BadDimension
                raise BadDimension("block_pointers must start at 0, end at the number of rows, "
                                   "ascend, and describe blocks of at most max_block_size rows")
            np_idx = np.int32 if a.row_ptrs.dtype == torch.int32 else np.int64
            self.block_pointers = ex.to_device(hp.astype(np_idx))
            self.num_blocks = self.block_pointers.numel() - 1
        else:
            self.block_pointers = ex.alloc((n + 1,), a.row_ptrs.dtype)
            nb = C.c_int64(0)
            call("gkoc_jacobi_find_blocks_" + self._suf, ex.stream, n,
                 a.row_ptrs, a.col_idxs, C.c_uint32(self.max_block_size),
                 C.byref(nb), self.block_pointers)
            self.num_blocks = nb.value
            self.block_pointers = self.block_pointers[:self.num_blocks + 1]
        gs = 1 << self.scheme.group_power
        storage = ((self.num_blocks + gs - 1) // gs) * self.scheme.group_offset
        # the inverse blocks are the apply's one large read stream: not with the vectors
        self.blocks = ex.zeros((storage,), a.dtype, MEM_INDICES)
        if adaptive:
            req = np.resize(np.asarray(block_wise, np.uint8), self.num_blocks)
            self.precisions = ex.to_device(req)
            self.conditioning = ex.alloc((self.num_blocks,), a.dtype)
            accuracy = C.c_double(factory.accuracy) if a.dtype == torch.float64 else C.c_float(factory.accuracy)
            call("gkoc_jacobi_generate_adaptive_" + self._suf, ex.stream, n,
                 a.row_ptrs, a.col_idxs, a.values, self.num_blocks, C.c_uint32(self.max_block_size),
                 self.scheme, self.block_pointers, accuracy, self.precisions,
                 self.conditioning, self.blocks)
            return
        call("gkoc_jacobi_generate_" + self._suf, ex.stream, n, a.row_ptrs,
             a.col_idxs, a.values, self.num_blocks,
             C.c_uint32(self.max_block_size), self.scheme, self.block_pointers,
             self.blocks, None)
        if self.storage_precision:
            call("gkoc_jacobi_convert_storage_f64", ex.stream, self.num_blocks, self.scheme,
                 self.blocks, C.c_uint8(self.storage_precision))

    def _apply_stored(self, alpha, b, beta, x):
        if self.precisions is not None:
            call("gkoc_jacobi_apply_adaptive_" + self._suf, self.exec.stream,
                 self.num_blocks, C.c_uint32(self.max_block_size), self.scheme, self.block_pointers,
                 self.blocks, self.precisions, None if alpha is None else alpha.values, b.values,
                 b.ld, None if beta is None else beta.values, x.values, x.ld, b.size[1])
            return
        call("gkoc_jacobi_apply_stored_f64_" + IT[self.block_pointers.dtype], self.exec.stream,
             self.num_blocks, C.c_uint32(self.max_block_size), self.scheme, self.block_pointers,
             self.blocks, C.c_uint8(self.storage_precision),
             None if alpha is None else alpha.values, b.values, b.ld,
             None if beta is None else beta.values, x.values, x.ld, b.size[1])

    def transpose(self):
        """Jacobi::transpose / conj_transpose (real types): every inverse block transposed,
        same storage scheme and precisions (jacobi::transpose_jacobi)"""
        import copy
        t = copy.copy(self)
        if self.max_block_size == 1:
            return t
        t.blocks = self.exec.zeros((self.blocks.numel(),), self.blocks.dtype)
        prec = self.precisions
        if prec is not None and self.dtype != torch.float64:
            call("gkoc_jacobi_transpose_adaptive_" + self._suf, self.exec.stream, self.num_blocks, self.scheme,
                 self.block_pointers, self.blocks, prec, C.c_int(0), t.blocks)
            return t
        if prec is None and self.storage_precision:
            prec = self.exec.to_device(np.full(self.num_blocks, self.storage_precision, np.uint8))
        call("gkoc_jacobi_transpose_" + self._suf, self.exec.stream, self.num_blocks,
             C.c_uint32(self.max_block_size), self.scheme, self.block_pointers, self.blocks, prec,
             t.blocks)
        return t

    conj_transpose = transpose

    def get_num_blocks(self):
        return self.num_blocks

    def apply_impl(self, b, x):
        ex = self.exec
        if self.max_block_size == 1:
            call("gkoc_jacobi_simple_scalar_apply_" + VT[self.dtype], ex.stream,
                 self.size[0], b.size[1], self.inv_diag, b.values, b.ld,
                 x.values, x.ld)
            return
        if self.storage_precision or self.precisions is not None:
            return self._apply_stored(None, b, None, x)
        call("gkoc_jacobi_simple_apply_" + self._suf, ex.stream,
             self.num_blocks, C.c_uint32(self.max_block_size), self.scheme,
             self.block_pointers, self.blocks, b.values, b.ld, x.values, x.ld,
             b.size[1])

    def can_fuse_dot(self, b):
        """x = M b together with <b, x> (gkoc_x_jacobi_simple_apply_dot_*): block
        storage with a power-of-two block_offset <= 16 and 64-wide groups, one
        right-hand side, unit stride"""
        if self.max_block_size == 1 or b.size[1] != 1 or b.ld != 1 or self.storage_precision \
                or self.precisions is not None:
            return False
        bo = self.scheme.block_offset
        return bo <= 16 and (bo & (bo - 1)) == 0 and (bo << self.scheme.group_power) == 64

    def apply_dot(self, b, x, dot_out, work):
        call("gkoc_x_jacobi_simple_apply_dot_" + self._suf, self.exec.stream,
             self.num_blocks, self.size[0], C.c_uint32(self.max_block_size), self.scheme,
             self.block_pointers, self.blocks, b.values, x.values, dot_out.values,
             work, C.c_size_t(work.numel() * work.element_size()))

    def can_fuse_step_2(self, b):
        """the fused step_2 + apply keeps two rows of per-workgroup partial sums in the workspace of
        gkoc_x_workspace_bytes: enough unless the blocks are tiny (< 2 rows on average)"""
        if not self.can_fuse_dot(b) or self.max_block_size == 1:
            return False
        from ._lib import lib
        return bool(lib().gkoc_x_cg_step_2_jacobi_apply_fits(
            C.c_int64(self.num_blocks), C.c_int64(self.size[0]), self.scheme,
            C.c_size_t(b.values.element_size())))

    def step_2_apply_dot(self, x, r, p, q, beta, rho, stop_status, z, rho_out, norm_out, take_sqrt,
                         work):
        """cg::step_2 (x += t p, r -= t q, t = rho / beta) and z = M r in one kernel, with
        rho_out = <r, z> and norm_out = ||r||^2 (||r|| with take_sqrt); same layouts as apply_dot
        (gkoc_x_cg_step_2_jacobi_apply_*).  x, r, z as from step_2 followed by apply, bit for bit."""
        call("gkoc_x_cg_step_2_jacobi_apply_" + self._suf, self.exec.stream, self.num_blocks,
             self.size[0], C.c_uint32(self.max_block_size), self.scheme, self.block_pointers,
             self.blocks, x.values, r.values, p.values, q.values, beta.values, rho.values,
             stop_status, z.values, rho_out.values, norm_out.values, C.c_int(1 if take_sqrt else 0),
             work, C.c_size_t(work.numel() * work.element_size()))

    def apply_advanced_impl(self, alpha, b, beta, x):
        ex = self.exec
        if self.max_block_size == 1:
            call("gkoc_jacobi_scalar_apply_" + VT[self.dtype], ex.stream,
                 self.size[0], b.size[1], self.inv_diag, alpha.values, b.values,
                 b.ld, beta.values, x.values, x.ld)
            return
        if self.storage_precision or self.precisions is not None:
            return self._apply_stored(alpha, b, beta, x)
        call("gkoc_jacobi_apply_" + self._suf, ex.stream, self.num_blocks,
             C.c_uint32(self.max_block_size), self.scheme, self.block_pointers,
             self.blocks, alpha.values, b.values, b.ld, beta.values, x.values,
             x.ld, b.size[1])
