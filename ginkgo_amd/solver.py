"""Krylov solvers on the Cdna4Executor.

Cg mirrors include/ginkgo/core/solver/cg.hpp and the driver
core/solver/cg.cpp:93-181 (Cg::apply_dense_impl): same factory interface
(`Cg.build().with_criteria(...).with_preconditioner(...).on(exec)
.generate(A)`), same kernel sequence per iteration (precond apply, conj_dot,
criterion check, step_1, SpMV, conj_dot, step_2, swap), same stopping
semantics.  Every step is a libgko_cdna4.so kernel; the host only drives.
"""
import ctypes as C

import torch

from ._lib import DimensionMismatch, NotSupported, VT, call
from .base import LinOp
from .matrix import Dense, scalar
from . import stop as _stop


class Identity(LinOp):
    """matrix::Identity - the default preconditioner."""

    def __init__(self, exec_, n):
        super().__init__(exec_, (n, n))

    def apply_impl(self, b, x):
        x.copy_from(b)

    def apply_advanced_impl(self, alpha, b, beta, x):
        x.scale(beta)
        x.add_scaled(alpha, b)


class _SolverFactory:
    def __init__(self, cls):
        self.cls = cls
        self.criteria = []
        self.preconditioner = None
        self.generated_preconditioner = None
        self.exec = None
        self.params = {}

    def with_criteria(self, *criteria):
        self.criteria = list(criteria)
        return self

    def with_preconditioner(self, factory):
        self.preconditioner = factory
        return self

    def with_generated_preconditioner(self, op):
        self.generated_preconditioner = op
        return self

    def __getattr__(self, name):
        if name.startswith("with_"):
            key = name[5:]

            def setter(v):
                self.params[key] = v
                return self
            return setter
        raise AttributeError(name)

    def on(self, exec_):
        self.exec = exec_
        for c in self.criteria:
            if c is not None and c.exec is None:
                c.on(exec_)
        if self.preconditioner is not None and self.preconditioner.exec is None:
            self.preconditioner.on(exec_)
        return self

    def generate(self, system_matrix):
        return self.cls(self, system_matrix)


class _IterativeSolver(LinOp):
    def __init__(self, factory, a):
        if a.size[0] != a.size[1]:
            raise DimensionMismatch("solver needs a square system matrix")
        super().__init__(factory.exec or a.exec, a.size)
        self.system_matrix = a
        self.criteria = factory.criteria
        if factory.generated_preconditioner is not None:
            self.preconditioner = factory.generated_preconditioner
        elif factory.preconditioner is not None:
            self.preconditioner = factory.preconditioner.generate(a)
        else:
            self.preconditioner = Identity(self.exec, a.size[0])
        self.params = dict(factory.params)
        # log::Convergence equivalent
        self.num_iterations = 0
        self.residual_norm = None
        self.has_converged = False
        self._ws = {}

    def get_system_matrix(self):
        return self.system_matrix

    def get_preconditioner(self):
        return self.preconditioner

    def _vec(self, name, like):
        v = self._ws.get(name)
        if v is None or v.size != like.size or v.dtype != like.dtype:
            v = Dense.create(self.exec, like.size, like.dtype)
            self._ws[name] = v
        return v

    def _scal(self, name, like):
        v = self._ws.get(name)
        if v is None or v.size != (1, like.size[1]) or v.dtype != like.dtype:
            v = Dense.create(self.exec, (1, like.size[1]), like.dtype)
            self._ws[name] = v
        return v

    def apply_advanced_impl(self, alpha, b, beta, x):
        # cg.cpp:184-200: x = beta*x + alpha*solve(b, x0 = x)
        xc = x.clone()
        self.apply_impl(b, xc)
        x.scale(beta)
        x.add_scaled(alpha, xc)


class Cg(_IterativeSolver):
    @staticmethod
    def build():
        return _SolverFactory(Cg)

    def apply_impl(self, b, x):
        ex = self.exec
        a, m = self.system_matrix, self.preconditioner
        suf = VT[b.dtype]
        rows, cols = b.size
        r, z = self._vec("r", b), self._vec("z", b)
        p, q = self._vec("p", b), self._vec("q", b)
        beta, prev_rho, rho = (self._scal(n, b) for n in ("beta", "prev_rho", "rho"))
        one = self._ws.setdefault(("one", b.dtype), scalar(ex, 1.0, b.dtype))
        neg_one = self._ws.setdefault(("neg", b.dtype), scalar(ex, -1.0, b.dtype))
        stop_status = self._ws.get("stop")
        if stop_status is None or stop_status.numel() != cols:
            stop_status = self._ws["stop"] = ex.zeros((cols,), torch.uint8)
        # r = b, z = p = q = 0, rho = 0, prev_rho = 1, stop.reset()
        call("gkoc_cg_initialize_" + suf, ex.stream, rows, cols, b.values, b.ld,
             r.values, r.ld, z.values, z.ld, p.values, p.ld, q.values, q.ld,
             prev_rho.values, rho.values, stop_status)
        # r = b - A x
        a.apply(neg_one, x, one, r)
        crit = _stop.combine(self.criteria, a, b, x, r)
        it = -1
        while True:
            m.apply(r, z)
            r.compute_conj_dot(z, rho)
            it += 1
            all_stopped, _ = crit.check(
                1, True, stop_status,
                {"num_iterations": it, "residual": r,
                 "implicit_sq_residual_norm": rho, "solution": x})
            if all_stopped:
                break
            call("gkoc_cg_step_1_" + suf, ex.stream, rows, cols, p.values, p.ld,
                 z.values, z.ld, rho.values, prev_rho.values, stop_status)
            a.apply(p, q)
            p.compute_conj_dot(q, beta)
            call("gkoc_cg_step_2_" + suf, ex.stream, rows, cols, x.values, x.ld,
                 r.values, r.ld, p.values, p.ld, q.values, q.ld, beta.values,
                 rho.values, stop_status)
            prev_rho, rho = rho, prev_rho
        self.num_iterations = it
        self.stop_status = stop_status
        st = stop_status.cpu()
        self.has_converged = bool(((st & 0x80) != 0).all().item())
        for c in crit.criteria:
            if getattr(c, "last_tau", None) is not None and not c.implicit:
                self.residual_norm = c.last_tau.to_numpy()[0]
