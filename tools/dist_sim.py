"""GPU-side cost of ONE rank of an N-rank z-slab run, on a single GPU: the
communicator is replaced by device copies of the right sizes (the values that
arrive are wrong, the kernels and their ordering are the real ones).  Gives the
per-iteration device time of the distributed CG without RCCL latency.
usage: python tools/dist_sim.py [grid=256] [world=8] [rank=3] [iters=200]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import ctypes as C

import ginkgo_amd as g
import ginkgo_amd.distributed as gd
from ginkgo_amd._lib import bump, call

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 3
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 200
plane = grid * grid
ex = g.Cdna4Executor.create(0)
part = gd.SlabPartition(grid, world)
lo, hi = part.range_of(rank)


class FakeComm:
    """every peer behaves like a translated copy of this rank"""
    host_staging = False

    def __init__(self):
        self.rank, self.size = rank, world
        # forks as the library's communicator makes them (csrc/comm.hip): two one-thread kernels,
        # or an event pair with GKOC_COMM_FORK=event
        self._fork_word = None if os.environ.get("GKOC_COMM_FORK") == "event" else \
            ex.zeros((64,), torch.int32)
        self._fork_n = C.c_uint32(0)

    def fork_deferred(self, side_stream):
        if self._fork_word is None:
            return None
        bump(self._fork_n)
        self._deferred = True
        return (self._fork_word, self._fork_n)

    def _fork(self, side_stream):
        if getattr(self, "_deferred", False):
            # the store is the product's first wave; only the poller here
            self._deferred = False
            call("gkoc_stream_fork_wait", C.c_void_p(side_stream.cuda_stream), self._fork_word, self._fork_n)
            return
        if self._fork_word is None:
            ev = torch.cuda.Event()
            ev.record()
            side_stream.wait_event(ev)
        else:
            bump(self._fork_n)
            call("gkoc_stream_fork", ex.stream, C.c_void_p(side_stream.cuda_stream), self._fork_word,
                 self._fork_n)

    def all_reduce_sum_(self, t):
        return t

    # the overlapped all-reduce: no transfer, but the same event records / cross-stream waits as
    # gkoc_comm_all_reduce_begin / _end (each is a barrier packet on the device)
    def all_reduce_begin(self, t, side_stream=None):
        if side_stream is not None:
            self._fork(side_stream)
            self._ar_done = torch.cuda.Event()
            self._ar_done.record(side_stream)
        return t

    def all_reduce_end(self):
        if getattr(self, "_ar_done", None) is not None:
            torch.cuda.current_stream().wait_event(self._ar_done)
            self._ar_done = None

    def all_to_all_counts(self, send_counts):
        return list(send_counts)

    # the device-resident exchange of RcclComm, with a device copy of the right size in place of
    # the transfer: on the side stream, ordered by events like gkoc_comm_exchange_begin / _join
    direct = True

    def exchange_begin(self, recv, send, recv_counts, send_counts, side_stream=None, send_displs=None):
        self._side = side_stream
        self._fork(side_stream)
        with torch.cuda.stream(side_stream):
            recv.view(-1).copy_(send.view(-1)[:recv.numel()])

    def exchange_end(self):
        self.exchange_join()

    if os.environ.get("GKO_SIM_SEPARATE_REDUCE") != "1":
        def all_reduce_exchange_begin(self, t, recv, send, recv_counts, send_counts, side_stream,
                                      send_displs=None):
            self.exchange_begin(recv, send, recv_counts, send_counts, side_stream, send_displs)

    def exchange_join(self):
        ev = torch.cuda.Event()
        ev.record(self._side)
        torch.cuda.current_stream().wait_event(ev)

    def exchange_forget(self):
        pass

    def all_to_all_v(self, recv, send, recv_counts, send_counts, async_op=False):
        if recv.dtype == torch.int64:
            # peers ask for the planes next to the ones we ask them for
            out, off = [], 0
            for p, c in enumerate(recv_counts):
                seg = send[off:off + c]
                out.append(seg + plane if p < rank else seg - plane)
                off += c
            recv.copy_(torch.cat(out) if out else send)
        else:
            recv.copy_(send)
        return None


z0, z1 = part.plane_offsets[rank], part.plane_offsets[rank + 1]
owned = g.stencil_csr(ex, 3, grid, z0=z0, nz=z1 - z0)
be = gd.HipBackend(ex)
a = gd.DistributedMatrix(be, FakeComm(), part, owned)
print(f"rank {rank}/{world} of {grid}^3: {hi-lo} rows, halo {a.n_halo} values in, {a.n_send} out")
x = be.vector_from(np.random.default_rng(1).uniform(-1, 1, hi - lo))
y = be.vector(hi - lo)
for _ in range(5):
    a.apply(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    a.apply(x, y)
e1.record()
torch.cuda.synchronize()
print(f"distributed SpMV (copy-exchange + complete boundary rows on the 2nd stream || interior rows): {e0.elapsed_time(e1)*20:.1f} us")
a.use_full_boundary = False
for _ in range(5):
    a.apply(x, y)
e0.record()
for _ in range(50):
    a.apply(x, y)
e1.record()
torch.cuda.synchronize()
print(f"distributed SpMV, round-2 path (exchange || whole local block, then boundary rows += halo part): {e0.elapsed_time(e1)*20:.1f} us")
a.use_full_boundary = os.environ.get("GKO_SIM_OLD_PATH") != "1"
for fused, s2, cls in ((False, False, gd.DistributedCg), (True, False, gd.DistributedCg),
                       (True, True, gd.DistributedCg), (True, False, gd.DistributedPipeCg),
                       (True, "steps", gd.DistributedPipeCg), (True, True, gd.DistributedPipeCg)):
    if os.environ.get("GKO_SIM_ONLY") == "pipe" and not (cls is gd.DistributedPipeCg and s2):
        continue
    if os.environ.get("GKO_SIM_ONLY") == "cg" and not (cls is gd.DistributedCg and s2):
        continue
    if os.environ.get("GKO_SIM_ONLY") == "x" and not s2:
        continue
    kw = dict(fused_step_2=s2) if cls is gd.DistributedCg else dict(fused_steps=bool(s2),
                                                                     fused_jacobi=s2 is True)
    s = cls(be, FakeComm(), a, iters, 1e-300, 8, fused=fused, **kw)
    rhs = be.vector_from(np.ones(hi - lo))
    xs = be.vector(hi - lo)
    s.apply(rhs, xs)
    torch.cuda.synchronize()
    xs.fill(0.0)
    t = time.perf_counter()
    s.apply(rhs, xs)
    torch.cuda.synchronize()
    t = time.perf_counter() - t
    print(f"{cls.__name__:18s} fused={fused!s:5s} step_2 fused with its neighbour={s2!s:5s}: {s.num_iterations} its, {t*1e6/max(s.num_iterations,1):8.1f} us/it "
          f"(device side, no RCCL latency) -> {max(s.num_iterations,1)/t:8.1f} it/s")

if os.environ.get("GKO_SIM_ONLY"):
    sys.exit(0)


# ---- pieces of the distributed SpMV
def tm(name, fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"   {name:46s} {e0.elapsed_time(e1)*1e3/reps:8.1f} us")


tm("local SpMV only (local columns)", lambda: be.spmv(a.local, x, y))
tm("pack (row_gather of the send planes)", lambda: be.gather(x, a.send_idx, a.send_buf))
tm("boundary rows (rowlist += halo part)", lambda: be.rowlist_add(a.nl, a.recv_buf, y))
if "full" in a.nl:
    r0, r1 = a.nl["full"]["interior"]
    tm("interior rows of the local block", lambda: be.spmv_rows(a.local, r0, r1, x, y))
    tm("complete boundary rows over [x | halo]", lambda: be.rowlist_full(a.nl, x, a.recv_buf, y))
tm("copy 'exchange' alone", lambda: a.recv_buf.values.copy_(a.send_buf.values))
tm("whole distributed apply", lambda: a.apply(x, y))
if a._gate is not None:
    xe = a.ext_vector()
    xe.values.copy_(x.values)
    tm("whole distributed apply, one-kernel product", lambda: a.apply(xe, y))
    y1 = y.values.clone()
    a.apply(x, y)
    torch.cuda.synchronize()
    print("   one-kernel product == join-based product bit for bit:", bool(torch.equal(y1, y.values)))
    a.check_gate()
