"""Stopping criteria - mirror of include/ginkgo/core/stop/{iteration,
residual_norm,combined}.hpp and core/stop/{iteration,residual_norm,
combined}.cpp.  stop_status is a device uint8 array with Ginkgo's bit layout
(stopping_status.hpp:117-121)."""
import ctypes as C

import torch

from ._lib import NotSupported, VT, call, cval
from .matrix import Dense


class mode:
    absolute = "absolute"
    initial_resnorm = "initial_resnorm"
    rhs_norm = "rhs_norm"


class _Factory:
    def __init__(self, cls, **params):
        self.cls, self.params, self.exec = cls, params, None

    def on(self, exec_):
        self.exec = exec_
        return self

    def generate(self, system_matrix, b, x, initial_residual):
        return self.cls(self, system_matrix, b, x, initial_residual)

    def __getattr__(self, name):
        if name.startswith("with_"):
            key = name[5:]

            def setter(v):
                self.params[key] = v
                return self
            return setter
        raise AttributeError(name)


class Iteration:
    """core/stop/iteration.cpp:15-26"""

    @staticmethod
    def build():
        return _Factory(Iteration, max_iters=0)

    def __init__(self, factory, a, b, x, r):
        self.max_iters = int(factory.params["max_iters"])
        self.exec = b.exec

    def check(self, stopping_id, set_finalized, stop_status, upd):
        hit = upd["num_iterations"] >= self.max_iters
        if hit:
            call("gkoc_set_all_statuses", self.exec.stream,
                 stop_status.numel(), C.c_uint8(stopping_id),
                 C.c_int(int(set_finalized)), stop_status)
        return hit, hit

    # deferred protocol (see Combined.check_begin): decided on the host at once
    def check_begin(self, stopping_id, set_finalized, stop_status, upd):
        return self.check(stopping_id, set_finalized, stop_status, upd)

    def check_done(self, token):
        return token


class ResidualNorm:
    """core/stop/residual_norm.cpp:75-205"""
    implicit = False

    @staticmethod
    def build():
        return _Factory(ResidualNorm, reduction_factor=5e-7,
                        baseline=mode.rhs_norm)

    def __init__(self, factory, a, b, x, r):
        ex = self.exec = b.exec
        self.reduction_factor = float(factory.params["reduction_factor"])
        baseline = self.baseline = factory.params["baseline"]
        cols = b.size[1]
        self.starting_tau = Dense.create(ex, (1, cols), b.dtype)
        if baseline == mode.rhs_norm:
            b.compute_norm2(self.starting_tau)
        elif baseline == mode.initial_resnorm:
            if r is None:
                raise NotSupported("initial_resnorm needs the initial residual")
            r.compute_norm2(self.starting_tau)
        elif baseline == mode.absolute:
            self.starting_tau.fill(1.0)
        else:
            raise NotSupported(f"unknown baseline {baseline}")
        self.u_dense_tau = Dense.create(ex, (1, cols), b.dtype)
        self.flags = ex.zeros((2,), torch.uint8)
        self.system = (a, b) if a is not None else None
        self._scratch = None

    def _select(self, upd):
        if upd.get("ignore_residual_check"):
            return None, None            # criterion::updater::ignore_residual_check
        if self.implicit:
            tau = upd.get("implicit_sq_residual_norm")
            if tau is None:
                raise NotSupported("ImplicitResidualNorm needs rho")
            name = "gkoc_implicit_residual_norm_"
        else:
            if upd.get("residual_norm") is not None:
                tau = upd["residual_norm"]
            elif upd.get("residual") is not None:
                upd["residual"].compute_norm2(self.u_dense_tau)
                tau = self.u_dense_tau
            elif upd.get("solution") is not None and self.system is not None:
                # no residual at hand: b - A x is computed (residual_norm.cpp:139-160)
                a, b = self.system
                if self._scratch is None or self._scratch.size != b.size:
                    from .matrix import Dense, scalar
                    self._scratch = Dense.create(self.exec, b.size, b.dtype)
                    self._one = scalar(self.exec, 1.0, b.dtype)
                    self._neg = scalar(self.exec, -1.0, b.dtype)
                self._scratch.copy_from(b)
                a.apply(self._neg, upd["solution"], self._one, self._scratch)
                self._scratch.compute_norm2(self.u_dense_tau)
                tau = self.u_dense_tau
            else:
                raise NotSupported("ResidualNorm needs a residual")
            name = "gkoc_residual_norm_"
        self.last_tau = tau
        return name, tau

    def check(self, stopping_id, set_finalized, stop_status, upd):
        name, tau = self._select(upd)
        if name is None:
            return False, False
        allc, chg = C.c_int(0), C.c_int(0)
        call(name + VT[tau.dtype], self.exec.stream, tau.size[1], tau.values,
             self.starting_tau.values, cval(tau.dtype, self.reduction_factor),
             C.c_uint8(stopping_id), C.c_int(int(set_finalized)), stop_status,
             self.flags, C.byref(allc), C.byref(chg))
        return bool(allc.value), bool(chg.value)

    _NSLOT = 16

    def check_begin(self, stopping_id, set_finalized, stop_status, upd):
        """enqueue the criterion kernel and a 2-byte copy of its flags into
        pinned host memory; nothing waits (gkoc_residual_norm_*, NULL host
        results).  check_done(token) blocks on exactly that copy."""
        name, tau = self._select(upd)
        if not hasattr(self, "_ring_dev"):
            self._ring_dev = self.exec.zeros((self._NSLOT, 2), torch.uint8)
            self._ring_host = torch.zeros((self._NSLOT, 2), dtype=torch.uint8).pin_memory()
            self._ring_next = 0
        slot = self._ring_next
        self._ring_next = (slot + 1) % self._NSLOT
        call(name + VT[tau.dtype], self.exec.stream, tau.size[1], tau.values,
             self.starting_tau.values, cval(tau.dtype, self.reduction_factor),
             C.c_uint8(stopping_id), C.c_int(int(set_finalized)), stop_status,
             self._ring_dev[slot], None, None)
        self._ring_host[slot].copy_(self._ring_dev[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return slot, ev

    def check_done(self, token):
        slot, ev = token
        ev.synchronize()
        h = self._ring_host[slot]
        return bool(h[0].item()), bool(h[1].item())


class ImplicitResidualNorm(ResidualNorm):
    """core/stop/residual_norm.cpp:209-230: sqrt(|rho|) <= factor * tau0"""
    implicit = True

    @staticmethod
    def build():
        return _Factory(ImplicitResidualNorm, reduction_factor=5e-7,
                        baseline=mode.rhs_norm)


class Combined:
    """core/stop/combined.cpp:33-51: criteria checked in order with ids 1,2,..."""

    def __init__(self, criteria):
        self.criteria = criteria

    def check(self, stopping_id, set_finalized, stop_status, upd):
        one_changed = False
        for i, c in enumerate(self.criteria):
            conv, chg = c.check(i + 1, set_finalized, stop_status, upd)
            one_changed |= chg
            if conv:
                return True, one_changed
        return False, one_changed

    def check_begin(self, stopping_id, set_finalized, stop_status, upd):
        """deferred form of check(): (tokens, decided).  Host-side criteria
        (Iteration) answer at once; `decided` is True when one of them stopped
        the solver, in which case later criteria are not evaluated (as in
        check()).  Device-side criteria only enqueue work."""
        tokens = []
        for i, c in enumerate(self.criteria):
            tok = c.check_begin(i + 1, set_finalized, stop_status, upd)
            tokens.append((c, tok))
            if isinstance(c, Iteration) and tok[0]:
                return tokens, True
        return tokens, False

    def check_done(self, tokens):
        one_changed = False
        for c, tok in tokens:
            conv, chg = c.check_done(tok)
            one_changed |= chg
            if conv:
                return True, one_changed
        return False, one_changed


def combine(factories, a, b, x, r):
    factories = [f for f in factories if f is not None]
    if not factories:
        raise NotSupported("no stopping criterion given")
    return Combined([f.generate(a, b, x, r) for f in factories])
