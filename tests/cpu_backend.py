"""TEST INFRASTRUCTURE: a CPU stand-in for ginkgo_amd.distributed.HipBackend built
on the oracle, so that the multi-rank logic (partition, halo plan, exchange,
all-reduce placement, CG driver) can be exercised under gloo without a GPU.
Never imported by the product."""
import numpy as np
import torch

from oracle import gko_oracle as o


class CpuVec:
    def __init__(self, arr):
        self.values = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).reshape(-1, 1))
        self.size = tuple(self.values.shape)
        self.dtype = torch.float64
        self.ld = 1

    def np(self):
        return self.values.numpy()[:, 0]

    def fill(self, v):
        self.values.fill_(v)
        return self

    def copy_from(self, other):
        self.values.copy_(other.values)
        return self

    def add_scaled(self, alpha, b):
        self.values.copy_(torch.from_numpy(
            o.dense_add_scaled(alpha.np(), b.np().copy(), self.np().copy()).reshape(-1, 1)))
        return self

    def to_numpy(self):
        return self.values.numpy().copy()


class CpuCsr:
    def __init__(self, n_rows, n_cols, rp, ci, v):
        self.size = (n_rows, n_cols)
        self.row_ptrs, self.col_idxs, self.values = rp, ci, v
        self.dtype = torch.float64


class OracleBackend:
    is_host = True

    def empty(self, n, dtype):
        return torch.empty(n, dtype=dtype)

    def zeros(self, n, dtype):
        return torch.zeros(n, dtype=dtype)

    def index_tensor(self, array, dtype):
        return torch.from_numpy(np.asarray(array)).to(dtype)

    def vector(self, n, dtype=torch.float64):
        return CpuVec(np.zeros(n))

    def vector_from(self, array):
        return CpuVec(array)

    def scalar(self, v, dtype=torch.float64):
        return CpuVec(np.array([v]))

    def to_host(self, t):
        return t.numpy()

    def split(self, a, lo, hi, n_global):
        rp, ci, v = a.row_ptrs, a.col_idxs, a.values
        n = len(rp) - 1
        is_local = (ci >= lo) & (ci < hi)
        row_of = np.repeat(np.arange(n), np.diff(rp))
        lrp = np.concatenate([[0], np.cumsum(np.bincount(row_of[is_local], minlength=n))]).astype(np.int32)
        local = CpuCsr(n, hi - lo, lrp, (ci[is_local] - lo).astype(np.int32), v[is_local])
        nl_cols_g = ci[~is_local]
        recv = np.unique(nl_cols_g)
        cnt = np.bincount(row_of[~is_local], minlength=n)
        rows = np.nonzero(cnt)[0].astype(np.int32)
        ptrs = np.concatenate([[0], np.cumsum(cnt[rows])]).astype(np.int32)
        nl = dict(rows=rows, ptrs=ptrs, cols=np.searchsorted(recv, nl_cols_g).astype(np.int32),
                  vals=v[~is_local], n=len(rows))
        return local, nl, torch.from_numpy(recv.astype(np.int32))

    def gather(self, x, idx, out):
        out.values.copy_(x.values[idx.long()])

    def spmv(self, a, x, y):
        y.values.copy_(torch.from_numpy(
            o.csr_spmv(a.row_ptrs, a.col_idxs, a.values, x.np().copy()).reshape(-1, 1)))

    def rowlist_add(self, nl, halo, y):
        yv, h = y.np(), halo.np()
        for i in range(nl["n"]):
            s = yv[nl["rows"][i]]
            for k in range(nl["ptrs"][i], nl["ptrs"][i + 1]):
                s += nl["vals"][k] * h[nl["cols"][k]]
            yv[nl["rows"][i]] = s

    def jacobi(self, a, max_block_size):
        be = self
        rp, ci, v = a.row_ptrs, a.col_idxs, a.values
        nb, ptrs = o.jacobi_find_blocks(rp, ci, max_block_size)
        scheme = o.jacobi_storage_scheme(max_block_size)
        blocks = o.jacobi_generate(rp, ci, v, nb, scheme, ptrs)

        class J:
            def apply(self, r, z):
                z.values.copy_(torch.from_numpy(
                    o.jacobi_apply(nb, scheme, ptrs, blocks, r.np().copy()).reshape(-1, 1)))
        return J()

    def cg_initialize(self, b, r, z, p, q, prev_rho, rho, stop):
        r.copy_from(b)
        for t in (z, p, q):
            t.fill(0.0)
        rho.fill(0.0)
        prev_rho.fill(1.0)
        stop.zero_()

    def cg_step_1(self, p, z, rho, prev_rho, stop):
        p.values.copy_(torch.from_numpy(o.cg_step_1(p.np().copy(), z.np().copy(), rho.np().copy(),
                                                    prev_rho.np().copy(), stop.numpy().copy())))

    def cg_step_2(self, x, r, p, q, beta, rho, stop):
        nx, nr = o.cg_step_2(x.np().copy(), r.np().copy(), p.np().copy(), q.np().copy(),
                             beta.np().copy(), rho.np().copy(), stop.numpy().copy())
        x.values.copy_(torch.from_numpy(nx))
        r.values.copy_(torch.from_numpy(nr))

    def local_dot(self, x, y, out):
        out.values[0, 0] = float(o.dense_dot(x.np().copy(), y.np().copy())[0])

    # pipe_cg step kernels: the oracle's restatement of reference/solver/pipe_cg_kernels.cpp,
    # working in place on the vectors' memory
    @staticmethod
    def _a(v):
        return v.values.numpy()

    def pipe_cg_initialize_1(self, b, r, prev_rho, stop):
        o.krylov_step("pipe_cg_initialize_1", b.size[0], 1, self._a(b), self._a(r), self._a(prev_rho),
                      stop.numpy())

    def pipe_cg_initialize_2(self, p, q, f, g, beta, z, w, m, n, delta):
        o.krylov_step("pipe_cg_initialize_2", p.size[0], 1, *(self._a(v) for v in
                                                              (p, q, f, g, beta, z, w, m, n, delta)))

    def pipe_cg_step_1(self, x, r, z, w, p, q, f, g, rho, beta, stop):
        o.krylov_step("pipe_cg_step_1", x.size[0], 1, *(self._a(v) for v in
                                                        (x, r, z, z, w, p, q, f, g, rho, beta)),
                      stop.numpy())

    def pipe_cg_step_2(self, beta, p, q, f, g, z, w, m, n, prev_rho, rho, delta, stop):
        o.krylov_step("pipe_cg_step_2", p.size[0], 1, *(self._a(v) for v in
                                                        (beta, p, q, f, g, z, w, m, n, prev_rho, rho,
                                                         delta)), stop.numpy())

    def scalar_tuple(self, k, dtype=torch.float64):
        t = torch.zeros(k, dtype=torch.float64)
        views = []
        for i in range(k):
            v = CpuVec.__new__(CpuVec)
            v.values = t[i:i + 1].view(1, 1)
            v.size, v.dtype, v.ld = (1, 1), torch.float64, 1
            views.append(v)
        return t, views

    def local_sqnorm(self, x, out):
        out.values[0, 0] = float(o.dense_norm2(x.np().copy(), squared=True)[0])

    def sqrt_(self, s):
        s.values.sqrt_()

    def stop_flags(self):
        return torch.zeros(2, dtype=torch.uint8), torch.zeros(1, dtype=torch.uint8)

    max_check_lag = 3

    def scalar_pair(self, dtype=torch.float64):
        t = torch.zeros(2, dtype=torch.float64)
        views = []
        for k in (0, 1):
            v = CpuVec.__new__(CpuVec)
            v.values = t[k:k + 1].view(1, 1)
            v.size, v.dtype, v.ld = (1, 1), torch.float64, 1
            views.append(v)
        return t, views[0], views[1]

    def check_begin(self, tau, tau0, factor, stop):
        return self.residual_check(tau, tau0, factor, stop, None)

    def check_done(self, token, block):
        return bool(token)

    def residual_check(self, tau, tau0, factor, stop, flags):
        allc, chg, st = o.residual_norm(tau.np().copy(), tau0.np().copy(), factor, 2, True,
                                        stop.numpy().copy())
        stop.copy_(torch.from_numpy(st))
        return allc

    def synchronize(self):
        pass
