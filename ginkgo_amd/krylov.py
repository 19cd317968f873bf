"""Bicgstab, Cgs, Fcg, PipeCg, Ir and Chebyshev on the Cdna4Executor.

Each class is the driver of the reference with the same kernel sequence -
  Bicgstab  core/solver/bicgstab.cpp:95-236
  Cgs       core/solver/cgs.cpp:96-201
  Fcg       core/solver/fcg.cpp:94-183
  PipeCg    core/solver/pipe_cg.cpp:95-297
  Bicg      core/solver/bicg.cpp:106-230
  Gcr       core/solver/gcr.cpp:95-320
  Minres    core/solver/minres.cpp:110-230
  Ir        core/solver/ir.cpp:189-255         (with core/solver/update_residual.hpp)
  Chebyshev core/solver/chebyshev.cpp:203-296  (likewise)
- issuing the fused vector updates of csrc/krylov_steps.hip (gkoc_bicgstab_*,
gkoc_cgs_*, gkoc_fcg_*, gkoc_pipe_cg_*) between the SpMV / preconditioner
applications and the reductions.  The criterion is checked where the reference
checks it (lock-step: these loops have two SpMVs per iteration, the host
round-trip of a check is small against them).
"""
import torch

from . import stop as _stop
from ._lib import VT, call
from .matrix import scalar
from .solver import _IterativeSolver, _SolverFactory


class _Krylov(_IterativeSolver):
    def _common(self, b):
        ex = self.exec
        one = self._ws.setdefault(("one", b.dtype), scalar(ex, 1.0, b.dtype))
        neg_one = self._ws.setdefault(("neg", b.dtype), scalar(ex, -1.0, b.dtype))
        cols = b.size[1]
        stop_status = self._ws.get("stop")
        if stop_status is None or stop_status.numel() != cols:
            stop_status = self._ws["stop"] = ex.zeros((cols,), torch.uint8)
        return ex, one, neg_one, stop_status

    def _check(self, crit, it, set_finalized, stop_status, residual, rho, x):
        upd = {"num_iterations": it, "residual": residual, "implicit_sq_residual_norm": rho}
        if x is not None:
            upd["solution"] = x
        return crit.check(1, set_finalized, stop_status, upd)

    def _finish(self, it, stop_status, r):
        self.num_iterations = it
        self.stop_status = stop_status
        self.has_converged = bool(((stop_status.cpu() & 0x80) != 0).all().item())
        tau = self._scal("report_norm", r)
        r.compute_norm2(tau)            # what log::Convergence reports
        self.residual_norm = tau.to_numpy()[0]


class Bicgstab(_Krylov):
    @staticmethod
    def build():
        return _SolverFactory(Bicgstab)

    def apply_impl(self, b, x):
        a, m = self.system_matrix, self.preconditioner
        ex, one, neg_one, stop = self._common(b)
        suf = VT[b.dtype]
        rows, cols = b.size
        r, z, y, v, s, t, p, rr = (self._vec(n, b) for n in ("r", "z", "y", "v", "s", "t", "p", "rr"))
        alpha, beta, gamma, prev_rho, rho, omega = (
            self._scal(n, b) for n in ("alpha", "beta", "gamma", "prev_rho", "rho", "omega"))
        st = lambda: ex.stream
        # r = b ; prev_rho = rho = omega = alpha = beta = gamma = 1 ; rr = v = s = t = z = y = p = 0
        call("gkoc_bicgstab_initialize_" + suf, st(), rows, cols, b.values, b.ld, r.values, r.ld,
             rr.values, rr.ld, y.values, y.ld, s.values, s.ld, t.values, t.ld, z.values, z.ld,
             v.values, v.ld, p.values, p.ld, prev_rho.values, rho.values, alpha.values,
             beta.values, gamma.values, omega.values, stop)
        a.apply(neg_one, x, one, r)                       # r = b - A x
        crit = _stop.combine(self.criteria, a, b, x, r)
        rr.copy_from(r)
        it = -1
        while True:
            it += 1
            rr.compute_conj_dot(r, rho)
            if self._check(crit, it, True, stop, r, rho, x)[0]:
                break
            # p = r + (rho / prev_rho * alpha / omega) (p - omega v)
            call("gkoc_bicgstab_step_1_" + suf, st(), rows, cols, r.values, r.ld, p.values, p.ld,
                 v.values, v.ld, rho.values, prev_rho.values, alpha.values, omega.values, stop)
            m.apply(p, y)
            a.apply(y, v)
            rr.compute_conj_dot(v, beta)
            # alpha = rho / beta ; s = r - alpha v
            call("gkoc_bicgstab_step_2_" + suf, st(), rows, cols, r.values, r.ld, s.values, s.ld,
                 v.values, v.ld, rho.values, alpha.values, beta.values, stop)
            all_stopped, one_changed = self._check(crit, it, False, stop, s, rho, None)
            if one_changed:
                # x += alpha y for the columns that just stopped
                call("gkoc_bicgstab_finalize_" + suf, st(), rows, cols, x.values, x.ld, y.values,
                     y.ld, alpha.values, stop)
            if all_stopped:
                break
            m.apply(s, z)
            a.apply(z, t)
            s.compute_conj_dot(t, gamma)
            t.compute_conj_dot(t, beta)
            # omega = gamma / beta ; x += alpha y + omega z ; r = s - omega t
            call("gkoc_bicgstab_step_3_" + suf, st(), rows, cols, x.values, x.ld, r.values, r.ld,
                 s.values, s.ld, t.values, t.ld, y.values, y.ld, z.values, z.ld, alpha.values,
                 beta.values, gamma.values, omega.values, stop)
            prev_rho, rho = rho, prev_rho
        self._finish(it, stop, r)


class Cgs(_Krylov):
    @staticmethod
    def build():
        return _SolverFactory(Cgs)

    def apply_impl(self, b, x):
        a, m = self.system_matrix, self.preconditioner
        ex, one, neg_one, stop = self._common(b)
        suf = VT[b.dtype]
        rows, cols = b.size
        r, r_tld, p, q, u, u_hat, v_hat, t = (
            self._vec(n, b) for n in ("r", "r_tld", "p", "q", "u", "u_hat", "v_hat", "t"))
        alpha, beta, gamma, prev_rho, rho = (
            self._scal(n, b) for n in ("alpha", "beta", "gamma", "prev_rho", "rho"))
        st = lambda: ex.stream
        call("gkoc_cgs_initialize_" + suf, st(), rows, cols, b.values, b.ld, r.values, r.ld,
             r_tld.values, r_tld.ld, p.values, p.ld, q.values, q.ld, u.values, u.ld,
             u_hat.values, u_hat.ld, v_hat.values, v_hat.ld, t.values, t.ld, alpha.values,
             beta.values, gamma.values, prev_rho.values, rho.values, stop)
        a.apply(neg_one, x, one, r)
        crit = _stop.combine(self.criteria, a, b, x, r)
        r_tld.copy_from(r)
        it = -1
        while True:
            r.compute_conj_dot(r_tld, rho)
            it += 1
            if self._check(crit, it, True, stop, r, rho, x)[0]:
                break
            # beta = rho / prev_rho ; u = r + beta q ; p = u + beta (q + beta p)
            call("gkoc_cgs_step_1_" + suf, st(), rows, cols, r.values, r.ld, u.values, u.ld,
                 p.values, p.ld, q.values, q.ld, beta.values, rho.values, prev_rho.values, stop)
            m.apply(p, t)
            a.apply(t, v_hat)
            r_tld.compute_conj_dot(v_hat, gamma)
            # alpha = rho / gamma ; q = u - alpha v_hat ; t = u + q
            call("gkoc_cgs_step_2_" + suf, st(), rows, cols, u.values, u.ld, v_hat.values,
                 v_hat.ld, q.values, q.ld, t.values, t.ld, alpha.values, rho.values,
                 gamma.values, stop)
            m.apply(t, u_hat)
            a.apply(u_hat, t)
            # r -= alpha t ; x += alpha u_hat
            call("gkoc_cgs_step_3_" + suf, st(), rows, cols, t.values, t.ld, u_hat.values,
                 u_hat.ld, r.values, r.ld, x.values, x.ld, alpha.values, stop)
            prev_rho, rho = rho, prev_rho
        self._finish(it, stop, r)


class Fcg(_Krylov):
    @staticmethod
    def build():
        return _SolverFactory(Fcg)

    def apply_impl(self, b, x):
        a, m = self.system_matrix, self.preconditioner
        ex, one, neg_one, stop = self._common(b)
        suf = VT[b.dtype]
        rows, cols = b.size
        r, z, p, q, t = (self._vec(n, b) for n in ("r", "z", "p", "q", "t"))
        beta, prev_rho, rho, rho_t = (self._scal(n, b) for n in ("beta", "prev_rho", "rho", "rho_t"))
        st = lambda: ex.stream
        # r = t = b ; rho = 0 ; prev_rho = rho_t = 1 ; z = p = q = 0
        call("gkoc_fcg_initialize_" + suf, st(), rows, cols, b.values, b.ld, r.values, r.ld,
             z.values, z.ld, p.values, p.ld, q.values, q.ld, t.values, t.ld, prev_rho.values,
             rho.values, rho_t.values, stop)
        a.apply(neg_one, x, one, r)
        crit = _stop.combine(self.criteria, a, b, x, r)
        it = -1
        while True:
            m.apply(r, z)
            r.compute_conj_dot(z, rho)
            t.compute_conj_dot(z, rho_t)
            it += 1
            if self._check(crit, it, True, stop, r, rho, x)[0]:
                break
            # p = z + (rho_t / prev_rho) p
            call("gkoc_fcg_step_1_" + suf, st(), rows, cols, p.values, p.ld, z.values, z.ld,
                 rho_t.values, prev_rho.values, stop)
            a.apply(p, q)
            p.compute_conj_dot(q, beta)
            # x += (rho / beta) p ; r -= (rho / beta) q ; t = r_new - r_old
            call("gkoc_fcg_step_2_" + suf, st(), rows, cols, x.values, x.ld, r.values, r.ld,
                 t.values, t.ld, p.values, p.ld, q.values, q.ld, beta.values, rho.values, stop)
            prev_rho, rho = rho, prev_rho
        self._finish(it, stop, r)


class PipeCg(_Krylov):
    """The reference interleaves (r, w) and (z1, z2) as the two columns of one
    matrix so that rho = <r, z> and delta = <w, z> come out of one 2-column dot
    (pipe_cg.cpp:106-160).  Here r, w, z are separate unit-stride vectors - the
    layout the SpMV and the preconditioner are fastest on - and z2 is z1 itself;
    the two dots are two reductions of the same values."""

    @staticmethod
    def build():
        return _SolverFactory(PipeCg)

    def apply_impl(self, b, x):
        a, m_op = self.system_matrix, self.preconditioner
        ex, one, neg_one, stop = self._common(b)
        suf = VT[b.dtype]
        rows, cols = b.size
        r, w, z, p, m, n, q, f, g = (
            self._vec(k, b) for k in ("r", "w", "z", "p", "m", "n", "q", "f", "g"))
        rho, delta, beta, prev_rho = (self._scal(k, b) for k in ("rho", "delta", "beta", "prev_rho"))
        st = lambda: ex.stream
        # r = b ; prev_rho = 1
        call("gkoc_pipe_cg_initialize_1_" + suf, st(), rows, cols, b.values, b.ld, r.values, r.ld,
             prev_rho.values, stop)
        a.apply(neg_one, x, one, r)
        m_op.apply(r, z)
        a.apply(z, w)
        m_op.apply(w, m)
        a.apply(m, n)
        r.compute_conj_dot(z, rho)
        w.compute_conj_dot(z, delta)
        crit = _stop.combine(self.criteria, a, b, x, r)
        it = 0
        if not self._check(crit, it, True, stop, r, rho, x)[0]:
            # beta = delta ; p = z ; q = w ; f = m ; g = n
            call("gkoc_pipe_cg_initialize_2_" + suf, st(), rows, cols, p.values, p.ld, q.values,
                 q.ld, f.values, f.ld, g.values, g.ld, beta.values, z.values, z.ld, w.values,
                 w.ld, m.values, m.ld, n.values, n.ld, delta.values)
            while True:
                # x += t p ; r -= t q ; z -= t f ; w -= t g   (t = rho / beta)
                call("gkoc_pipe_cg_step_1_" + suf, st(), rows, cols, x.values, x.ld, r.values,
                     r.ld, z.values, z.ld, z.values, z.ld, w.values, w.ld, p.values, p.ld,
                     q.values, q.ld, f.values, f.ld, g.values, g.ld, rho.values, beta.values,
                     stop)
                m_op.apply(w, m)
                a.apply(m, n)
                prev_rho.copy_from(rho)
                r.compute_conj_dot(z, rho)
                w.compute_conj_dot(z, delta)
                it += 1
                if self._check(crit, it, True, stop, r, rho, x)[0]:
                    break
                # beta = delta - |rho / prev_rho|^2 beta ; p = z + t p ; q = w + t q ; ...
                call("gkoc_pipe_cg_step_2_" + suf, st(), rows, cols, beta.values, p.values, p.ld,
                     q.values, q.ld, f.values, f.ld, g.values, g.ld, z.values, z.ld, w.values,
                     w.ld, m.values, m.ld, n.values, n.ld, prev_rho.values, rho.values,
                     delta.values, stop)
        self._finish(it, stop, r)


class Bicg(_Krylov):
    """Biconjugate gradients (core/solver/bicg.cpp:106-230): the system and its transposed
    shadow advance together; A^T and M^T come from Csr.transpose / Jacobi.transpose, built
    at every apply like in the reference."""

    @staticmethod
    def build():
        return _SolverFactory(Bicg)

    def apply_impl(self, b, x):
        a, m = self.system_matrix, self.preconditioner
        ex, one, neg_one, stop = self._common(b)
        suf = VT[b.dtype]
        rows, cols = b.size
        r, z, p, q, r2, z2, p2, q2 = (self._vec(n, b) for n in ("r", "z", "p", "q", "r2", "z2", "p2", "q2"))
        beta, prev_rho, rho = (self._scal(n, b) for n in ("beta", "prev_rho", "rho"))
        st = lambda: ex.stream
        call("gkoc_bicg_initialize_" + suf, st(), rows, cols, b.values, b.ld, r.values, r.ld,
             z.values, z.ld, p.values, p.ld, q.values, q.ld, prev_rho.values, rho.values,
             r2.values, r2.ld, z2.values, z2.ld, p2.values, p2.ld, q2.values, q2.ld, stop)
        at, mt = a.transpose(), m.transpose()
        a.apply(neg_one, x, one, r)
        r2.copy_from(r)
        crit = _stop.combine(self.criteria, a, b, x, r)
        it = -1
        while True:
            m.apply(r, z)
            mt.apply(r2, z2)
            z.compute_conj_dot(r2, rho)
            it += 1
            if self._check(crit, it, True, stop, r, rho, x)[0]:
                break
            # p = z + (rho / prev_rho) p ; p2 = z2 + (rho / prev_rho) p2
            call("gkoc_bicg_step_1_" + suf, st(), rows, cols, p.values, p.ld, z.values, z.ld,
                 p2.values, p2.ld, z2.values, z2.ld, rho.values, prev_rho.values, stop)
            a.apply(p, q)
            at.apply(p2, q2)
            p2.compute_conj_dot(q, beta)
            # x += t p ; r -= t q ; r2 -= t q2   (t = rho / beta)
            call("gkoc_bicg_step_2_" + suf, st(), rows, cols, x.values, x.ld, r.values, r.ld,
                 r2.values, r2.ld, p.values, p.ld, q.values, q.ld, q2.values, q2.ld, beta.values,
                 rho.values, stop)
            prev_rho, rho = rho, prev_rho
        self._finish(it, stop, r)


class Minres(_Krylov):
    """MINRES for symmetric, possibly indefinite systems (core/solver/minres.cpp:110-230):
    preconditioned Lanczos recurrence + Givens rotations.  The driver hands no residual to
    the criterion, only tau = ||z||^2 (ImplicitResidualNorm uses it; a plain ResidualNorm
    recomputes b - A x, as in the reference)."""

    @staticmethod
    def build():
        return _SolverFactory(Minres)

    def apply_impl(self, b, x):
        a, m = self.system_matrix, self.preconditioner
        ex, one, neg_one, stop = self._common(b)
        suf = VT[b.dtype]
        rows, cols = b.size
        r, z, p, q, v, z_tilde, p_prev, q_prev = (
            self._vec(k, b) for k in ("r", "z", "p", "q", "v", "z_tilde", "p_prev", "q_prev"))
        (alpha, beta, gamma, delta, eta_next, eta, tau, cos_prev, cos, sin_prev, sin) = (
            self._scal(k, b) for k in ("alpha", "beta", "gamma", "delta", "eta_next", "eta", "tau",
                                       "cos_prev", "cos", "sin_prev", "sin"))
        st = lambda: ex.stream
        r.copy_from(b)
        a.apply(neg_one, x, one, r)
        crit = _stop.combine(self.criteria, a, b, x, r)
        m.apply(r, z)
        r.compute_conj_dot(z, beta)
        z.compute_conj_dot(z, tau)
        # (v takes the place of q_tilde, minres.cpp:169-178)
        call("gkoc_minres_initialize_" + suf, st(), rows, cols, r.values, r.ld, z.values, z.ld,
             p.values, p.ld, p_prev.values, p_prev.ld, q.values, q.ld, q_prev.values, q_prev.ld,
             v.values, v.ld, beta.values, gamma.values, delta.values, cos_prev.values, cos.values,
             sin_prev.values, sin.values, eta_next.values, eta.values, stop)
        it = -1
        while True:
            it += 1
            if crit.check(1, True, stop, {"num_iterations": it, "implicit_sq_residual_norm": tau,
                                          "solution": x})[0]:
                break
            a.apply(one, z, neg_one, v)                   # v = A z - v
            v.compute_conj_dot(z, alpha)
            v.sub_scaled(alpha, q)
            m.apply(v, z_tilde)
            v.compute_conj_dot(z_tilde, beta)
            call("gkoc_minres_step_1_" + suf, st(), cols, alpha.values, beta.values, gamma.values,
                 delta.values, cos_prev.values, cos.values, sin_prev.values, sin.values, eta.values,
                 eta_next.values, tau.values, stop)
            p, p_prev = p_prev, p
            call("gkoc_minres_step_2_" + suf, st(), rows, cols, x.values, x.ld, p.values, p.ld,
                 p_prev.values, p_prev.ld, z.values, z.ld, z_tilde.values, z_tilde.ld, q.values,
                 q.ld, q_prev.values, q_prev.ld, v.values, v.ld, alpha.values, beta.values,
                 gamma.values, delta.values, cos.values, eta.values, stop)
            gamma, beta = beta, gamma
        self._finish(it, stop, r)      # (r is the initial residual, as the logger sees it)


class Gcr(_Krylov):
    """Restarted generalised conjugate residual method (core/solver/gcr.cpp:95-320):
    search directions p_k and A p_k kept in two tall matrices of krylov_dim + 1 slots,
    A p_k orthogonalised against the earlier ones with modified Gram-Schmidt."""

    @staticmethod
    def build():
        return _SolverFactory(Gcr)

    def apply_impl(self, b, x):
        from .matrix import Dense
        a, m = self.system_matrix, self.preconditioner
        ex, one, neg_one, stop = self._common(b)
        suf = VT[b.dtype]
        n, cols = b.size
        kd = int(self.params.get("krylov_dim", 100)) or 100
        r, pr, apr = (self._vec(k, b) for k in ("residual", "precon_residual", "A_precon_residual"))
        key = ("gcr", n, cols, kd, b.dtype)
        if self._ws.get("gcr_key") != key:
            self._ws["gcr_key"] = key
            self._ws["P"] = Dense.create(ex, ((kd + 1) * n, cols), b.dtype)
            self._ws["AP"] = Dense.create(ex, ((kd + 1) * n, cols), b.dtype)
            self._ws["ap_norms"] = Dense.create(ex, (kd + 1, cols), b.dtype)
            self._ws["final"] = ex.zeros((cols,), torch.int64)
        P, AP, ap_norms, final = (self._ws[k] for k in ("P", "AP", "ap_norms", "final"))
        rap, minus_beta, rnorm = (self._scal(k, b) for k in ("rAp", "minus_beta", "residual_norm"))
        slot = lambda mat, i: mat.create_submatrix((n * i, n * (i + 1)), (0, cols))
        st = lambda: ex.stream

        def restart():
            call("gkoc_gcr_restart_" + suf, st(), n, cols, pr.values, pr.ld, apr.values, apr.ld,
                 P.values, P.ld, AP.values, AP.ld, final)

        call("gkoc_gcr_initialize_" + suf, st(), n, cols, b.values, b.ld, r.values, r.ld, stop)
        a.apply(neg_one, x, one, r)
        m.apply(r, pr)
        a.apply(pr, apr)
        restart()
        crit = _stop.combine(self.criteria, a, b, x, r)
        it, restart_iter = -1, 0
        while True:
            it += 1
            r.compute_norm2(rnorm)
            if crit.check(1, True, stop, {"num_iterations": it, "residual": r, "residual_norm": rnorm,
                                          "solution": x})[0]:
                break
            if restart_iter == kd:
                restart()
                restart_iter = 0
            Ap, p = slot(AP, restart_iter), slot(P, restart_iter)
            r.compute_conj_dot(Ap, rap)
            ap_norm = ap_norms.create_submatrix((restart_iter, restart_iter + 1), (0, cols))
            Ap.compute_squared_norm2(ap_norm)
            # x += t p ; r -= t Ap   (t = <r, Ap> / ||Ap||^2)
            call("gkoc_gcr_step_1_" + suf, st(), n, cols, x.values, x.ld, r.values, r.ld, p.values,
                 p.ld, Ap.values, Ap.ld, ap_norm.values, rap.values, stop)
            m.apply(r, pr)
            a.apply(pr, apr)
            next_Ap, next_p = slot(AP, restart_iter + 1), slot(P, restart_iter + 1)
            next_Ap.copy_from(apr)
            next_p.copy_from(pr)
            for i in range(restart_iter + 1):
                Ap_i, p_i = slot(AP, i), slot(P, i)
                apr.compute_conj_dot(Ap_i, minus_beta)
                minus_beta.inv_scale(ap_norms.create_submatrix((i, i + 1), (0, cols)))
                next_Ap.sub_scaled(minus_beta, Ap_i)
                next_p.sub_scaled(minus_beta, p_i)
            restart_iter += 1
        self._finish(it, stop, r)


class _Stationary(_Krylov):
    """residual handling shared by Ir and Chebyshev (core/solver/update_residual.hpp:25-75):
    iteration 0 checks the initial residual; later iterations first ask the criteria that
    need no residual, then recompute r = b - A x and check it.  The initial guess is x."""

    def _update_residual(self, crit, it, a, b, x, r, one, neg_one, stop):
        if it == 0:
            return crit.check(1, True, stop, {"num_iterations": it, "residual": r, "solution": x})[0]
        if crit.check(1, False, stop, {"num_iterations": it, "solution": x,
                                       "ignore_residual_check": True})[0]:
            return True
        r.copy_from(b)
        a.apply(neg_one, x, one, r)
        return crit.check(1, True, stop, {"num_iterations": it, "residual": r, "solution": x})[0]

    def _finish_stationary(self, it, stop, a, b, x, r, one, neg_one):
        # what log::Convergence reports: the residual of the final x
        r.copy_from(b)
        a.apply(neg_one, x, one, r)
        self._finish(it, stop, r)


class Ir(_Stationary):
    """x += relaxation_factor * S (b - A x) with inner solver S (with_solver /
    with_preconditioner, Identity if none): Richardson iteration / iterative refinement"""

    @staticmethod
    def build():
        return _SolverFactory(Ir)

    def apply_impl(self, b, x):
        a, inner = self.system_matrix, self.preconditioner
        ex, one, neg_one, stop = self._common(b)
        r = self._vec("r", b)
        relax = scalar(ex, float(self.params.get("relaxation_factor", 1.0)), b.dtype)
        call("gkoc_ir_initialize", ex.stream, b.size[1], stop)
        r.copy_from(b)
        a.apply(neg_one, x, one, r)
        crit = _stop.combine(self.criteria, a, b, x, r)
        it = -1
        while True:
            it += 1
            if self._update_residual(crit, it, a, b, x, r, one, neg_one, stop):
                break
            inner.apply(relax, r, one, x)          # x = relaxation * S r + x
        self._finish_stationary(it, stop, a, b, x, r, one, neg_one)


class Chebyshev(_Stationary):
    """Chebyshev iteration for a preconditioned operator whose spectrum lies between the
    foci (with_foci((lower, upper)))"""

    @staticmethod
    def build():
        return _SolverFactory(Chebyshev)

    def apply_impl(self, b, x):
        import ctypes as C
        a, m = self.system_matrix, self.preconditioner
        ex, one, neg_one, stop = self._common(b)
        suf = VT[b.dtype]
        rows, cols = b.size
        r, inner, update = (self._vec(n, b) for n in ("r", "inner_solution", "update_solution"))
        lo, hi = self.params.get("foci", (0.0, 1.0))
        center, direction = (lo + hi) / 2.0, (hi - lo) / 2.0
        if center == 0:
            raise ValueError("Chebyshev: the centre of the foci must not be zero")
        alpha = 1.0 / center
        beta = 0.5 * (direction * alpha) * (direction * alpha)
        call("gkoc_ir_initialize", ex.stream, cols, stop)
        r.copy_from(b)
        a.apply(neg_one, x, one, r)
        crit = _stop.combine(self.criteria, a, b, x, r)
        it = -1
        while True:
            it += 1
            if self._update_residual(crit, it, a, b, x, r, one, neg_one, stop):
                break
            m.apply(r, inner)
            if it == 0:
                # update = inner ; x += alpha inner
                call("gkoc_chebyshev_init_update_" + suf, ex.stream, rows, cols, C.c_double(alpha),
                     inner.values, inner.ld, update.values, update.ld, x.values, x.ld)
                continue
            if it > 1:
                beta = (direction * alpha / 2.0) * (direction * alpha / 2.0)
            alpha = 1.0 / (center - beta / alpha)
            # inner = update = inner + beta update ; x += alpha inner
            call("gkoc_chebyshev_update_" + suf, ex.stream, rows, cols, C.c_double(alpha),
                 C.c_double(beta), inner.values, inner.ld, update.values, update.ld, x.values, x.ld)
        self._finish_stationary(it, stop, a, b, x, r, one, neg_one)
