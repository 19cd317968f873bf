"""Krylov solvers on the Cdna4Executor.

Cg mirrors include/ginkgo/core/solver/cg.hpp and the driver
core/solver/cg.cpp:93-181 (Cg::apply_dense_impl): same factory interface
(`Cg.build().with_criteria(...).with_preconditioner(...).on(exec)
.generate(A)`), same kernel sequence per iteration (precond apply, conj_dot,
criterion check, step_1, SpMV, conj_dot, step_2, swap), same stopping
semantics.  Every step is a libgko_cdna4.so kernel; the host only drives.
"""
import ctypes as C
import os

from collections import deque

import torch

from ._lib import DimensionMismatch, NotSupported, VT, call, lib
from .base import LinOp
from .executor import MEM_VALUES, MEM_VECTOR
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

    def transpose(self):
        return self

    conj_transpose = transpose


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
            # Memory class of a workspace vector (arena): a kernel's output should
            # not share a class with its large inputs.  z is written by the preconditioner, whose
            # inputs are the Jacobi blocks (index class) and r (vector class): z goes to the
            # third class, where the matrix values live (no kernel touches values and z).
            role = MEM_VALUES if name == "z" and os.environ.get("GKO_CG_Z_ROLE", "1") == "1" \
                else MEM_VECTOR
            v = Dense.create(self.exec, like.size, like.dtype, role=role)
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
        # The criterion is evaluated every iteration as in cg.cpp:133-141, but
        # its answer is read `check_lag` iterations later: the kernel marks
        # stop_status on the device and step_1 / step_2 skip stopped columns, so
        # the iterations enqueued in between leave x, r, p as they were at the
        # stopping iteration.  with_check_lag(0) = lock-step like the reference.
        lag = int(self.params.get("check_lag", 4))
        # the criterion's flag ring holds _NSLOT checks in flight: a larger lag would overwrite
        # slots that have not been read yet
        lag = max(0, min(lag, _stop.ResidualNorm._NSLOT - 2))
        # Fused producer+reduction kernels (include/gko_cdna4.h, gkoc_x_*): same
        # vectors bit for bit, one pass less over r / z / p / q per pair.
        # with_fused_kernels(False) = the reference's kernel sequence.
        fuse = bool(self.params.get("fused_kernels", True)) and cols == 1 and \
            all(v.ld == 1 for v in (b, x, r, z, p, q))
        from .matrix import Csr
        from .preconditioner import Jacobi
        # spmv + <p, q> in one pass: 1034 against 1050 us on L256 once the operands sit in
        # different memory classes (round 1, without the arena: slower fused); with_fused_spmv_dot(False)
        # turns it off
        fuse_spmv = fuse and isinstance(a, Csr) and bool(self.params.get("fused_spmv_dot", True))
        fuse_prec = fuse and isinstance(m, Jacobi) and m.can_fuse_dot(r)
        fuse_norm = fuse and any(isinstance(c, _stop.ResidualNorm) and not c.implicit
                                 for c in crit.criteria)
        work = None
        if fuse_spmv or fuse_prec or fuse_norm:
            nbytes = lib().gkoc_x_workspace_bytes(C.c_int64(rows), C.c_size_t(b.values.element_size()))
            work = self._ws.get("fused_work")
            if work is None or work.numel() * work.element_size() < nbytes:
                work = self._ws["fused_work"] = ex.alloc(
                    ((nbytes + b.values.element_size() - 1) // b.values.element_size(),), b.dtype)
        tau = self._scal("tau", b) if fuse_norm else None
        have_tau = False
        pending = deque()
        it = -1
        # step_2 of iteration k and the preconditioner application of iteration k+1 in ONE kernel
        # (gkoc_x_cg_step_2_jacobi_apply_*): the new residual goes from the registers that computed
        # it into the block product; x, r, z, the iteration count and the stop status are those of
        # the separate kernels bit for bit.  with_fused_step_2_apply(False) turns it off.
        fuse_s2 = fuse_prec and fuse_norm and m.can_fuse_step_2(r) and \
            bool(self.params.get("fused_step_2_apply", True))
        have_z = False     # z and rho of the coming iteration are already there

        def precond_and_rho(rho_):
            if fuse_prec:
                m.apply_dot(r, z, rho_, work)
            else:
                m.apply(r, z)
                r.compute_conj_dot(z, rho_)

        def update(rho_, prev_rho_):
            call("gkoc_cg_step_1_" + suf, ex.stream, rows, cols, p.values, p.ld,
                 z.values, z.ld, rho_.values, prev_rho_.values, stop_status)
            if fuse_spmv:
                a.apply_dot(p, q, beta, work)
            else:
                a.apply(p, q)
                p.compute_conj_dot(q, beta)
            if fuse_s2 and not in_graph[0]:
                # also z = M r_new and the NEXT iteration's rho, written over prev_rho
                # (step_1 above was its last reader)
                m.step_2_apply_dot(x, r, p, q, beta, rho_, stop_status, z, prev_rho_, tau, True, work)
            elif fuse_norm:
                call("gkoc_x_cg_step_2_norm_" + suf, ex.stream, rows, x.values, r.values,
                     p.values, q.values, beta.values, rho_.values, stop_status, tau.values,
                     C.c_int(1), work, C.c_size_t(work.numel() * work.element_size()))
            else:
                call("gkoc_cg_step_2_" + suf, ex.stream, rows, cols, x.values, x.ld,
                     r.values, r.ld, p.values, p.ld, q.values, q.ld, beta.values,
                     rho_.values, stop_status)

        # hipGraph mode (with_hip_graph(True), or "auto" = systems below 2^21 rows,
        # where the loop is bound by host launch cost): two consecutive
        # iterations - the rho / prev_rho roles swap back after two - are captured
        # once and replayed; the criterion kernels inside the graph leave their
        # flags in two device slots that are copied to pinned memory after each
        # replay and read `check_lag` iterations later, exactly like the eager path.
        self._graph_x_ptr = x.values.data_ptr()
        graph = self._graph_setup(crit, fuse_norm, lag, rows, cols) if fuse_norm else None
        in_graph = [graph is not None]     # the captured iterations keep the two-kernel sequence
        if graph is not None:
            fuse_s2 = False
        while True:
            if graph is not None and have_tau and it + 2 < graph["max_iters"]:
                graph["exec"] = self._graph_capture(
                    graph, lambda rho_, prev_rho_: precond_and_rho(rho_),
                    update, rho, prev_rho, tau, stop_status)
                graph["exec"].replay()
                stopped = None
                for k in (0, 1):
                    it += 1
                    pending.append((it, [(graph["reader"], graph["reader"].adopt(graph["flags"][k]))]))
                while pending and pending[0][0] <= it - lag:
                    pit, ptok = pending.popleft()
                    if crit.check_done(ptok)[0]:
                        stopped = pit
                        break
                if stopped is not None:
                    it = stopped
                    break
                continue
            if have_z:
                pass                                   # the fused step_2 left z and rho
            elif fuse_prec:
                m.apply_dot(r, z, rho, work)
            else:
                m.apply(r, z)
                r.compute_conj_dot(z, rho)
            it += 1
            tokens, decided = crit.check_begin(
                1, True, stop_status,
                {"num_iterations": it, "residual": r,
                 "residual_norm": tau if have_tau else None,
                 "implicit_sq_residual_norm": rho, "solution": x})
            pending.append((it, tokens))
            stopped = None
            while pending and (decided or pending[0][0] <= it - lag):
                pit, ptok = pending.popleft()
                if crit.check_done(ptok)[0]:
                    stopped = pit
                    break
            if stopped is not None:
                it = stopped
                break
            update(rho, prev_rho)
            have_tau = fuse_norm
            have_z = fuse_s2
            prev_rho, rho = rho, prev_rho
        self.num_iterations = it
        self.stop_status = stop_status
        st = stop_status.cpu()
        self.has_converged = bool(((st & 0x80) != 0).all().item())
        # log::Convergence semantics (core/log/convergence.cpp): the residual norm of the iterate
        # the solver stopped at.  With the fused step_2 + norm kernels tau already holds ||r||
        # (a stopped column leaves r alone, so the run-ahead iterations recompute the same value);
        # otherwise - the iteration limit hit before ResidualNorm looked at this r - it is computed
        if have_tau:
            self.residual_norm = tau.to_numpy()[0]
        else:
            fin = self._scal("final_norm", b)
            r.compute_norm2(fin)
            self.residual_norm = fin.to_numpy()[0]


    # ------------------------------------------------------------ hipGraph
    class _GraphFlags:
        """reads the flags a captured criterion kernel left on the device: same
        token protocol as stop.ResidualNorm.check_begin / check_done"""
        _NSLOT = 32

        def __init__(self, exec_):
            self.host = torch.zeros((self._NSLOT, 2), dtype=torch.uint8).pin_memory()
            self.next = 0

        def adopt(self, dev_flags):
            slot = self.next
            self.next = (slot + 1) % self._NSLOT
            self.host[slot].copy_(dev_flags, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            return slot, ev

        def check_done(self, token):
            slot, ev = token
            ev.synchronize()
            return bool(self.host[slot, 0].item()), bool(self.host[slot, 1].item())

    def _graph_setup(self, crit, fuse_norm, lag, rows, cols):
        """None unless graph mode applies: one right-hand side, fused norm, criteria
        made of Iteration and one ResidualNorm, lag >= 1"""
        mode = self.params.get("hip_graph", "auto")
        if mode is False or not fuse_norm or lag < 1 or cols != 1:
            return None
        if mode == "auto" and rows >= (1 << 21):
            return None
        rn, max_iters, rn_id = None, float("inf"), 0
        for i, c in enumerate(crit.criteria):
            if isinstance(c, _stop.Iteration):
                max_iters = min(max_iters, c.max_iters)
            elif type(c) is _stop.ResidualNorm and rn is None:
                rn, rn_id = c, i + 1
            else:
                return None
        if rn is None:
            return None
        g_ = self._ws.get("graph")
        if g_ is None:
            g_ = self._ws["graph"] = {
                "flags": self.exec.zeros((2, 2), torch.uint8),
                "tau0": Dense.create(self.exec, (1, 1), rn.starting_tau.dtype),
                "reader": Cg._GraphFlags(self.exec), "exec": None, "key": None}
        g_["tau0"].copy_from(rn.starting_tau)      # criteria are rebuilt per apply
        g_.update(max_iters=max_iters, rn_id=rn_id, factor=rn.reduction_factor)
        return g_

    def _graph_capture(self, g_, precond_and_rho, update, rho, prev_rho, tau, stop_status):
        key = (rho.values.data_ptr(), prev_rho.values.data_ptr(), tau.values.data_ptr(),
               stop_status.data_ptr(), g_["rn_id"], g_["factor"],
               tuple(v.values.data_ptr() for v in (self._ws["r"], self._ws["z"], self._ws["p"],
                                                   self._ws["q"])), self._graph_x_ptr)
        if g_["key"] == key and g_["exec"] is not None:
            return g_["exec"]
        ex = self.exec
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=ex.device)
        side.wait_stream(torch.cuda.current_stream(ex.device))
        with torch.cuda.graph(graph, stream=side):
            for k, (rk, pk) in enumerate(((rho, prev_rho), (prev_rho, rho))):
                precond_and_rho(rk, pk)
                call("gkoc_residual_norm_" + VT[tau.dtype], ex.stream, 1, tau.values,
                     g_["tau0"].values, _stop.cval(tau.dtype, g_["factor"]),
                     C.c_uint8(g_["rn_id"]), C.c_int(1), stop_status, g_["flags"][k], None, None)
                update(rk, pk)
        torch.cuda.current_stream(ex.device).wait_stream(side)
        g_["key"] = key
        return graph


class ortho_method:
    """include/ginkgo/core/solver/gmres.hpp: gmres::ortho_method"""
    mgs = "mgs"
    cgs = "cgs"
    cgs2 = "cgs2"


class Gmres(_IterativeSolver):
    """Restarted GMRES - mirror of include/ginkgo/core/solver/gmres.hpp and the
    driver core/solver/gmres.cpp:321-621 (apply_dense_impl): same workspace
    layout (Krylov basis = one tall Dense of (krylov_dim+1)*n rows, Hessenberg
    stored row-per-iteration), same kernel sequence, MGS / CGS / CGS2
    orthogonalisation (:157-300), Givens QR and restart logic.  Parameters:
    with_krylov_dim (default 100), with_ortho_method, with_flexible."""

    @staticmethod
    def build():
        return _SolverFactory(Gmres)

    # ---- reduction hooks: identity on one GPU; distributed.DistributedGmres
    # all-reduces the device-resident values over the row partition
    _comm = None

    def _reduce(self, t):
        if self._comm is not None:
            self._comm.all_reduce_sum_(t)

    def _norm2(self, v, out):
        if self._comm is None:
            v.compute_norm2(out)
        else:
            v.compute_squared_norm2(out)
            self._reduce(out.values.view(-1))
            call("gkoc_dense_compute_sqrt_" + VT[out.dtype], self.exec.stream, out.size[1], out.values)

    def _dot(self, v, w, out):
        v.compute_conj_dot(w, out)
        self._reduce(out.values.view(-1))

    def _residual(self, a, b_like_residual, x, one, neg_one, scratch):
        """residual (holding b) -= A x"""
        if self._comm is None:
            a.apply(neg_one, x, one, b_like_residual)
        else:
            a.apply(x, scratch)
            b_like_residual.sub_scaled(one, scratch)

    def apply_impl(self, b, x):
        import ctypes as C
        from . import _lib
        ex = self.exec
        a, m = self.system_matrix, self.preconditioner
        suf = VT[b.dtype]
        n, nrhs = b.size
        kd = int(self.params.get("krylov_dim", 100)) or 100
        ortho = self.params.get("ortho_method", ortho_method.mgs)
        flexible = bool(self.params.get("flexible", False))
        dt = b.dtype
        D = lambda shape: Dense.create(ex, shape, dt)
        residual, precv = self._vec("residual", b), self._vec("precv", b)
        before, after = self._vec("before", b), self._vec("after", b)
        key = ("gmres", n, nrhs, kd, dt, flexible)
        if self._ws.get("gmres_key") != key:
            self._ws["gmres_key"] = key
            self._ws["krylov"] = D(((kd + 1) * n, nrhs))
            self._ws["pkrylov"] = D(((kd + 1) * n, nrhs)) if flexible else None
            self._ws["hess"] = D((kd, (kd + 1) * nrhs))
            self._ws["haux"] = D((kd + 1, nrhs))
            self._ws["gsin"], self._ws["gcos"] = D((kd, nrhs)), D((kd, nrhs))
            self._ws["rnc"] = D((kd + 1, nrhs))
            self._ws["rnorm"] = D((1, nrhs))
            self._ws["y"] = D((kd, nrhs))
            self._ws["final"] = ex.zeros((nrhs,), torch.int64)   # size_type
            self._ws["gstop"] = ex.zeros((nrhs,), torch.uint8)
            need = _lib.lib().gkoc_gmres_multi_dot_workspace_bytes
            need.restype = C.c_size_t
            self._ws["mdwork"] = ex.alloc(
                (max(need(C.c_int64(n), C.c_int64(nrhs), C.c_int64(kd + 1),
                          C.c_size_t(8 if dt == torch.float64 else 4)), 8),), torch.uint8)
        w = self._ws
        krylov, pkrylov, hess, haux = w["krylov"], w["pkrylov"], w["hess"], w["haux"]
        gsin, gcos, rnc, rnorm, y = w["gsin"], w["gcos"], w["rnc"], w["rnorm"], w["y"]
        final, stop_status, mdwork = w["final"], w["gstop"], w["mdwork"]
        one = w.setdefault(("one", dt), scalar(ex, 1.0, dt))
        neg_one = w.setdefault(("neg", dt), scalar(ex, -1.0, dt))

        def basis(mat, i):
            return mat.create_submatrix((n * i, n * (i + 1)), (0, nrhs))

        def restart():
            call("gkoc_gmres_restart_" + suf, ex.stream, n, nrhs, residual.values, residual.ld,
                 rnorm.values, rnc.values, krylov.values, krylov.ld, final)

        def multi_dot(next_k, num, target):
            call("gkoc_gmres_multi_dot_" + suf, ex.stream, n, nrhs, num, krylov.values, krylov.ld,
                 next_k.values, next_k.ld, target.values, target.ld, mdwork,
                 C.c_size_t(mdwork.numel()))
            self._reduce(target.values[:num])

        call("gkoc_common_gmres_initialize_" + suf, ex.stream, n, nrhs, b.values, b.ld,
             residual.values, residual.ld, gsin.values, gsin.ld, gcos.values, gcos.ld, kd,
             stop_status)
        self._residual(a, residual, x, one, neg_one, before)
        self._norm2(residual, rnorm)
        restart()
        crit = _stop.combine(self.criteria, a, b, x, residual)
        if self._comm is not None:
            for c in crit.criteria:           # baselines are norms of distributed vectors
                if isinstance(c, _stop.ResidualNorm) and c.baseline != _stop.mode.absolute:
                    self._norm2(b if c.baseline == _stop.mode.rhs_norm else residual, c.starting_tau)
        total_iter, restart_iter = -1, 0
        while True:
            total_iter += 1
            all_stopped, _ = crit.check(
                1, False, stop_status,
                {"num_iterations": total_iter, "residual": residual, "residual_norm": rnorm,
                 "solution": x})
            if all_stopped:
                break
            if restart_iter == kd:
                call("gkoc_common_gmres_solve_krylov_" + suf, ex.stream, nrhs, rnc.values, rnc.ld,
                     hess.values, hess.ld, y.values, y.ld, final, stop_status)
                call("gkoc_gmres_multi_axpy_" + suf, ex.stream, n, nrhs, krylov.values, krylov.ld,
                     y.values, y.ld, before.values, before.ld, final, stop_status)
                m.apply(before, after)
                x.add_scaled(one, after)
                residual.copy_from(b)
                self._residual(a, residual, x, one, neg_one, before)
                self._norm2(residual, rnorm)
                restart()
                restart_iter = 0
            this_k, next_k = basis(krylov, restart_iter), basis(krylov, restart_iter + 1)
            pre_k = basis(pkrylov, restart_iter) if flexible else precv
            m.apply(this_k, pre_k)
            # hessenberg_iter: (restart_iter + 2) x nrhs view into row restart_iter
            hrow = hess.values[restart_iter, :(restart_iter + 2) * nrhs]
            hiter = Dense(ex, hrow.view(restart_iter + 2, nrhs))
            a.apply(pre_k, next_k)
            if ortho == ortho_method.mgs:
                mgs_fused = bool(self.params.get("fused_kernels", True)) and nrhs == 1 and \
                    krylov.ld == 1
                if mgs_fused:
                    # h_0 = <v_0, w>; then each step w -= h_i v_i together with
                    # h_{i+1} = <v_{i+1}, w>: one pass over w less per step
                    xw = self._ws.get("x_work")
                    need = _lib.lib().gkoc_x_workspace_bytes(C.c_int64(n), C.c_size_t(b.values.element_size()))
                    if xw is None or xw.numel() * xw.element_size() < need:
                        es = b.values.element_size()
                        xw = self._ws["x_work"] = ex.alloc(((need + es - 1) // es,), b.dtype)
                    h0 = hiter.create_submatrix((0, 1), (0, 1))
                    self._dot(basis(krylov, 0), next_k, h0)
                    for i in range(restart_iter):
                        call("gkoc_x_gmres_mgs_step_" + suf, ex.stream, n, next_k.values,
                             basis(krylov, i).values, hiter.values[i:i + 1],
                             basis(krylov, i + 1).values, hiter.values[i + 1:i + 2], xw,
                             C.c_size_t(xw.numel() * xw.element_size()))
                        self._reduce(hiter.values[i + 1:i + 2].view(-1))
                    last = restart_iter
                    next_k.sub_scaled(hiter.create_submatrix((last, last + 1), (0, 1)),
                                      basis(krylov, last))
                else:
                    for i in range(restart_iter + 1):
                        h_i = hiter.create_submatrix((i, i + 1), (0, nrhs))
                        self._dot(basis(krylov, i), next_k, h_i)
                        next_k.sub_scaled(h_i, basis(krylov, i))
            else:
                fused = bool(self.params.get("fused_kernels", True))

                def subtract(hm):
                    # next_k -= sum_i hm(i,:) * basis_i: one pass (bit-identical to
                    # the restart_iter + 1 sub_scaled calls of the reference)
                    if fused:
                        call("gkoc_x_gmres_multi_sub_scaled_" + suf, ex.stream, n, nrhs,
                             restart_iter + 1, krylov.values, krylov.ld, hm.values, hm.ld,
                             next_k.values, next_k.ld)
                    else:
                        for i in range(restart_iter + 1):
                            next_k.sub_scaled(hm.create_submatrix((i, i + 1), (0, nrhs)),
                                              basis(krylov, i))
                multi_dot(next_k, restart_iter + 1, hiter)
                subtract(hiter)
                if ortho == ortho_method.cgs2:
                    aux = haux.create_submatrix((0, restart_iter + 2), (0, nrhs))
                    multi_dot(next_k, restart_iter + 1, aux)
                    subtract(aux)
                    hiter.add_scaled(one, aux)
            h_norm = hiter.create_submatrix((restart_iter + 1, restart_iter + 2), (0, nrhs))
            self._norm2(next_k, h_norm)
            next_k.inv_scale(h_norm)
            call("gkoc_common_gmres_hessenberg_qr_" + suf, ex.stream, nrhs, gsin.values, gsin.ld,
                 gcos.values, gcos.ld, rnorm.values, rnc.values, rnc.ld, hiter.values, hiter.ld,
                 restart_iter, final, stop_status)
            restart_iter += 1
        call("gkoc_common_gmres_solve_krylov_" + suf, ex.stream, nrhs, rnc.values, rnc.ld,
             hess.values, hess.ld, y.values, y.ld, final, stop_status)
        if flexible:
            call("gkoc_gmres_multi_axpy_" + suf, ex.stream, n, nrhs, pkrylov.values, pkrylov.ld,
                 y.values, y.ld, after.values, after.ld, final, stop_status)
        else:
            call("gkoc_gmres_multi_axpy_" + suf, ex.stream, n, nrhs, krylov.values, krylov.ld,
                 y.values, y.ld, before.values, before.ld, final, stop_status)
            m.apply(before, after)
        x.add_scaled(one, after)
        self.num_iterations = total_iter
        self.stop_status = stop_status
        self.has_converged = bool(((stop_status.cpu() & 0x80) != 0).all().item())
        self.residual_norm = rnorm.to_numpy()[0]
