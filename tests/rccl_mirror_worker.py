"""One rank, backend "nccl" (= RCCL): the collectives of the distributed path go
through the real RCCL calls on the device (all_reduce on 2-value device tensors,
all_to_all_single with split sizes on the side stream), with the mirror-rank trick
of test_overlap_branch_with_mirror_comm supplying the second slab of a z-symmetric
problem.  What a single GPU cannot show is the xGMI transport; the torch.distributed
call pattern, dtypes, stream ordering and buffer reuse under RCCL it can."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def mirror_problem(grid, direct=False):
    """rank 0 of a 2-slab run whose peer is its own mirror image; collectives via RCCL:
    through torch.distributed (TorchComm) or, direct=True, through the C ABI (RcclComm)"""
    import ctypes as C
    import ginkgo_amd as g
    import ginkgo_amd.distributed as gd
    from ginkgo_amd._lib import call

    plane = grid * grid
    calls = {"ar": 0, "a2a": 0}
    ex = g.Cdna4Executor.create(0)

    class DirectMirrorComm(gd.RcclComm):
        def __init__(self):
            super().__init__(ex)               # a real 1-rank RCCL communicator
            self.rank, self.size = 0, 2
            self.two = g.scalar(ex, 2.0, torch.float64)

        def all_reduce_sum_(self, t):
            # the peer's contribution equals ours: sum = 2 x.  Everything through call()
            # so that the solver's call tape (_lib.Tape) replays it.
            assert t.is_cuda and t.dtype == torch.float64
            call("gkoc_comm_all_reduce_sum", self._handle, self.exec.stream, t, t.numel(),
                 C.c_size_t(t.element_size()))
            call("gkoc_dense_scale_f64", self.exec.stream, t.numel(), 1, self.two.values, 1, t, 1)
            calls["ar"] += 1
            return t

        def all_reduce_begin(self, t, side_stream=None):
            # the overlapped form (PipeCg): the real RCCL all-reduce on the side stream ...
            side = C.c_void_p(side_stream.cuda_stream) if side_stream is not None else None
            call("gkoc_comm_all_reduce_begin", self._handle, self.exec.stream, side, t, t.numel(),
                 C.c_size_t(t.element_size()))
            self._pending = t
            calls["ar_overlapped"] = calls.get("ar_overlapped", 0) + 1
            return t

        def all_reduce_end(self):
            # ... and the peer's equal contribution once the main stream has the result
            call("gkoc_comm_all_reduce_end", self._handle, self.exec.stream)
            t = self._pending
            call("gkoc_dense_scale_f64", self.exec.stream, t.numel(), 1, self.two.values, 1, t, 1)

        def all_to_all_counts(self, send_counts):
            return list(send_counts)

        def all_reduce_exchange_begin(self, t, recv, send, recv_counts, send_counts, side_stream,
                                      send_displs=None):
            # reduction + exchange behind one fork: the real combined RCCL call (to ourselves) ...
            assert list(recv_counts) == [0, plane] and list(send_counts) == [0, plane]
            calls["ar_with_exchange"] = calls.get("ar_with_exchange", 0) + 1
            super().all_reduce_exchange_begin(t, recv, send, [plane], [plane], side_stream,
                                              None if send_displs is None else [send_displs[1]])
            # ... and the peer's equal contribution, on the exchange's stream right behind the
            # reduction: whoever waits for that stream - a join, or the step kernel through the
            # product's gate (no join at all) - finds the doubled values
            call("gkoc_dense_scale_f64", C.c_void_p(side_stream.cuda_stream), t.numel(), 1, self.two.values, 1,
                 t, 1)

        def exchange_begin(self, recv, send, recv_counts, send_counts, side_stream=None,
                           send_displs=None):
            assert list(recv_counts) == [0, plane] and list(send_counts) == [0, plane]
            calls["a2a"] += 1
            calls["zero_copy"] = calls.get("zero_copy", 0) + (send_displs is not None)
            super().exchange_begin(recv, send, [plane], [plane], side_stream,     # to ourselves
                                   None if send_displs is None else [send_displs[1]])

        def all_to_all_v(self, recv, send, recv_counts, send_counts, async_op=False):
            assert recv.dtype == torch.int64   # set-up only; floats go through exchange_begin
            recv.copy_(send - plane)
            return None

    class RcclMirrorComm(gd.TorchComm):
        def __init__(self):
            super().__init__(None)
            assert not self.host_staging
            self.rank, self.size = 0, 2

        def all_reduce_sum_(self, t):
            dist.all_reduce(t)                 # world 1: identity, through RCCL
            calls["ar"] += 1
            return t.mul_(2)

        def all_to_all_counts(self, send_counts):
            return list(send_counts)

        def all_to_all_v(self, recv, send, recv_counts, send_counts, async_op=False):
            assert list(recv_counts) == [0, plane] and list(send_counts) == [0, plane]
            dist.all_to_all_single(recv, send, [plane], [plane])
            calls["a2a"] += 1
            if recv.dtype == torch.int64:
                recv.sub_(plane)
            return None

    part = gd.SlabPartition(grid, 2)
    owned = g.stencil_csr(ex, 3, grid, z0=0, nz=grid // 2)
    be = gd.HipBackend(ex)
    comm = DirectMirrorComm() if direct else RcclMirrorComm()
    a = gd.DistributedMatrix(be, comm, part, owned)
    assert a._side is not None and a.n_halo == plane
    a._owned_for_tests = owned
    return be, comm, a, part, calls


def ex_stream(be):
    return be.exec.stream


def C_void(stream):
    import ctypes as C
    return C.c_void_p(stream.cuda_stream)


def C_size(v):
    import ctypes as C
    return C.c_size_t(v)


def raw_all_reduce(handle, stream, t):
    """gkoc_comm_all_reduce_sum without the raising wrapper: the return code"""
    import ctypes as C
    from ginkgo_amd._lib import lib
    f = lib().gkoc_comm_all_reduce_sum
    f.restype = C.c_int
    return f(handle, stream, C.c_void_p(t.data_ptr()), C.c_int64(t.numel()), C.c_size_t(t.element_size()))


def init_rccl_single():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))


def main():
    import ginkgo_amd.distributed as gd
    from oracle import gko_oracle as oracle

    init_rccl_single()
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    direct = len(sys.argv) > 2 and sys.argv[2] == "direct"
    plane, n = grid * grid, grid ** 3
    be, comm, a, part, calls = mirror_problem(grid, direct)
    lo, hi = part.range_of(0)
    rp, ci, v = oracle.stencil_csr(3, grid)
    half = np.random.default_rng(9).uniform(-1, 1, n // 2)
    xg = np.concatenate([half, half.reshape(grid // 2, plane)[::-1].reshape(-1)])
    x, y = be.vector_from(xg[lo:hi]), be.vector(hi - lo)
    for _ in range(5):
        a.apply(x, y)
    ref = oracle.csr_spmv(rp, ci, v, xg)[lo:hi]
    got = y.to_numpy()[:, 0]
    assert np.max(np.abs(got - ref)) <= 1e-14 * np.max(np.abs(ref))
    solver = gd.DistributedCg(be, comm, a, 500, 1e-10, 8)
    xs = be.vector(hi - lo)
    solver.apply(be.vector_from(np.ones(hi - lo)), xs)
    xo, iters, _ = oracle.cg_solve(rp, ci, v, np.ones(n), max_iters=500, reduction=1e-10, precond="block")
    assert abs(solver.num_iterations - iters) <= 1, (solver.num_iterations, iters)
    e = np.linalg.norm(xs.to_numpy()[:, 0] - xo[lo:hi]) / np.linalg.norm(xo[lo:hi])
    assert e < 1e-8, e
    gm = gd.DistributedGmres(be, comm, a, 400, 1e-9, 8, krylov_dim=12, ortho_method="cgs")
    xg_ = be.vector(hi - lo)
    gm.apply(be.vector_from(np.ones(hi - lo)), xg_)
    xo2, it2, _ = oracle.gmres_solve(rp, ci, v, np.ones(n), krylov_dim=12, ortho="cgs", max_iters=400,
                                     reduction=1e-9, precond="block", max_block_size=8)
    # restarted GMRES counts drift with rounding on long runs: exact only on the small grid
    assert gm.has_converged and abs(gm.num_iterations - it2) <= max(1, it2 // 10), (gm.num_iterations, it2)
    # distributed PipeCg: one (overlapped) all-reduce per iteration
    pipe = gd.DistributedPipeCg(be, comm, a, 500, 1e-10, 8)
    xq = be.vector(hi - lo)
    pipe.apply(be.vector_from(np.ones(hi - lo)), xq)
    xo3, it3, _ = oracle.krylov_solve("pipe_cg", rp, ci, v, np.ones(n), max_iters=500, reduction=1e-10,
                                      precond="block")
    assert abs(pipe.num_iterations - it3) <= 1, (pipe.num_iterations, it3)
    e = np.linalg.norm(xq.to_numpy()[:, 0] - xo3[lo:hi]) / np.linalg.norm(xo3[lo:hi])
    assert e < 1e-8, e
    if direct:
        import ctypes as C  # noqa: F401
        from ginkgo_amd._lib import call
        # default: the reduction starts together with the halo exchange behind one fork (opened by
        # the product's first wave), and nothing joins: the step kernel waits for the product's gate
        # and evaluates the criterion itself ...
        assert pipe.taped and calls.get("ar_with_exchange", 0) >= 2, calls
        joined = gd.DistributedPipeCg(be, comm, a, 500, 1e-10, 8)
        joined.step_gate = False               # ... the join-based form gives the same bits
        xj = be.vector(hi - lo)
        joined.apply(be.vector_from(np.ones(hi - lo)), xj)
        assert joined.num_iterations == pipe.num_iterations
        assert np.array_equal(xj.to_numpy(), xq.to_numpy())
        # ... without the preconditioner inside the step kernel m is not final at that point:
        # all_reduce_begin / _end on the side stream, the exchange inside the SpMV
        sep = gd.DistributedPipeCg(be, comm, a, 500, 1e-10, 8, fused_jacobi=False)
        xs_ = be.vector(hi - lo)
        sep.apply(be.vector_from(np.ones(hi - lo)), xs_)
        assert calls.get("ar_overlapped", 0) >= 2 and sep.num_iterations == pipe.num_iterations, calls
        # (the three sums come from another reduction tree there, so the iterates agree to rounding)
        d = np.linalg.norm(xs_.to_numpy() - xq.to_numpy()) / np.linalg.norm(xq.to_numpy())
        assert d < 1e-9, d
        plain_pipe = gd.DistributedPipeCg(be, comm, a, 500, 1e-10, 8, taped=False, check_lag=0)
        xq0 = be.vector(hi - lo)
        plain_pipe.apply(be.vector_from(np.ones(hi - lo)), xq0)
        assert plain_pipe.num_iterations == pipe.num_iterations
        assert np.array_equal(xq0.to_numpy(), xq.to_numpy())
        # an all-reduce on the main stream while an overlapped one is pending must be refused
        t2 = torch.ones(2, dtype=torch.float64, device="cuda")
        comm_h = comm._handle
        call("gkoc_comm_all_reduce_begin", comm_h, ex_stream(be), C_void(a._side), t2, 2, C_size(8))
        rc = raw_all_reduce(comm_h, ex_stream(be), t2)
        call("gkoc_comm_all_reduce_end", comm_h, ex_stream(be))
        assert rc != 0, "all-reduce on another stream while one is pending was accepted"
    if direct:
        # what bench.py does before it times anything at N > 1: the one-kernel product against the
        # join-based one on this communicator; and the way out it takes when that fails
        ok, why = a.self_check()
        assert ok and "bit for bit" in why, why
        a2 = gd.DistributedMatrix(be, comm, part, a._owned_for_tests)
        a2.conservative()
        assert a2._gate is None and getattr(a2.ext_vector(), "_ext_halo", None) is None
        cons = gd.DistributedCg(be, comm, a2, 500, 1e-10, 8)
        xc = be.vector(hi - lo)
        cons.apply(be.vector_from(np.ones(hi - lo)), xc)
        assert cons.num_iterations == solver.num_iterations
        assert np.array_equal(xc.to_numpy(), xs.to_numpy())
    if direct:
        # the recorded-call loop (Tape) against the plain loop: same bits
        assert solver.taped and comm.tapeable and calls["ar"] > 4
        assert calls["zero_copy"] > 0 and a.send_displs == [0, (grid // 2 - 1) * plane]
        plain = gd.DistributedCg(be, comm, a, 500, 1e-10, 8, taped=False)
        xp = be.vector(hi - lo)
        plain.apply(be.vector_from(np.ones(hi - lo)), xp)
        assert plain.num_iterations == solver.num_iterations
        assert np.array_equal(xp.to_numpy(), xs.to_numpy())
    else:
        assert calls["ar"] > 2 * iters and calls["a2a"] > iters
    # timing of the pieces under RCCL (informational)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(50):
        a.apply(x, y)
    torch.cuda.synchronize()
    t_apply = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(50):
        comm.all_reduce_sum_(y.values[:2].view(-1))
    torch.cuda.synchronize()
    t_ar = (time.perf_counter() - t0) / 50
    print(f"rccl_mirror OK {'direct' if direct else 'torch'} grid={grid} cg_iters={solver.num_iterations} all_reduce_calls={calls['ar']} "
          f"a2a_calls={calls['a2a']} apply_us={t_apply * 1e6:.1f} all_reduce_us={t_ar * 1e6:.1f}")
    if direct:
        comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
