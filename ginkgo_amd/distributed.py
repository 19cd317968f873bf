"""Row-partitioned matrices / vectors / CG across GPUs (one process per GPU).

Mirror of gko::experimental::distributed::{Partition, Matrix, Vector} and of the
distributed Cg path for the hot configuration:
  include/ginkgo/core/distributed/partition.hpp:229-262 (contiguous partitions),
  core/distributed/matrix.cpp:300-381 (read_distributed: local / non-local split,
  index_map, RowGatherer set-up) and :450-509 (apply: pack halo -> exchange ||
  local SpMV -> non-local SpMV), core/distributed/vector.cpp:473-592 (dot / norm
  = local kernel + all-reduce).

MI355X-native differences: the exchange runs over RCCL (torch.distributed
backend "nccl") on device buffers - no MPI, no host staging -, asynchronously
to the local SpMV; the non-local part is a row list so only boundary rows are
touched; the matrix is generated and split on the device.

`backend` supplies the numerical kernels.  The product backend is HipBackend
(libgko_cdna4.so); tests inject a CPU backend to exercise the communication
logic under gloo.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from ._lib import IT, VT, GkoError, bump, call, lib
from ._lib import record as _record
from .executor import MEM_INDICES, MEM_VALUES
from .matrix import Csr, Dense, scalar, stencil_csr
from .preconditioner import Jacobi


IPC_HANDLE_BYTES = 128        # GKOC_COMM_IPC_HANDLE_BYTES
BUS_ID_BYTES = 32             # GKOC_COMM_BUS_ID_BYTES


class CommTopology(C.Structure):
    """gkoc_comm_topology (include/gko_cdna4.h)"""
    _fields_ = [("transport", C.c_int32), ("n_ranks", C.c_int32), ("rank", C.c_int32),
                ("ranks_seen", C.c_int32), ("rccl_version", C.c_int32), ("cross_device", C.c_int32),
                ("window_uncached", C.c_int32), ("gate_fence", C.c_int32),
                ("bus_id", (C.c_char * BUS_ID_BYTES) * 16)]


def gate_fence_policy(set_to=-1):
    """gkoc_gate_fence_policy: what the kernels that read behind a gate word pay per waiting wave (0 cheap
    gate, 1 agent-scope acquire, 2 system-scope acquire); returns the policy in force afterwards"""
    now = C.c_int(0)
    call("gkoc_gate_fence_policy", C.c_int(int(set_to)), C.byref(now))
    return int(now.value)


class StepGate(C.Structure):
    """gkoc_step_gate (include/gko_cdna4.h): what a fused PipeCg step kernel waits for and the
    stopping criterion it evaluates itself"""
    _fields_ = [("wait_word", C.c_void_p), ("wait_number", C.c_uint32), ("implicit", C.c_int32),
                ("tau", C.c_void_p), ("orig_tau", C.c_void_p), ("goal", C.c_double),
                ("flags", C.c_void_p), ("stopping_id", C.c_uint8), ("set_finalized", C.c_uint8)]


class Partition:
    """Contiguous 1-D row partition; `offsets` has n_parts + 1 entries."""

    def __init__(self, offsets):
        self.offsets = [int(o) for o in offsets]
        if any(b < a for a, b in zip(self.offsets, self.offsets[1:])) or self.offsets[0] != 0:
            raise GkoError("Partition: offsets must start at 0 and be non-decreasing")

    @property
    def n_parts(self):
        return len(self.offsets) - 1

    @property
    def n_global(self):
        return self.offsets[-1]

    def range_of(self, part):
        return self.offsets[part], self.offsets[part + 1]

    def owner_of(self, gidx):
        """owning part of each global index (numpy array)"""
        return np.searchsorted(np.asarray(self.offsets[1:]), gidx, side="right")

    @staticmethod
    def build_from_global_size_uniform(n_parts, n_global):
        """partition.hpp:262: the first n_global % n_parts parts get one more"""
        base, rest = divmod(n_global, n_parts)
        off = [0]
        for p in range(n_parts):
            off.append(off[-1] + base + (1 if p < rest else 0))
        return Partition(off)

    @staticmethod
    def build_slabs(grid, n_parts, nd=3):
        """plane-aligned z-slabs (y-rows for nd = 2) of a grid^nd stencil"""
        planes = Partition.build_from_global_size_uniform(n_parts, grid)
        plane_size = grid ** (nd - 1)
        return Partition([o * plane_size for o in planes.offsets])


class TorchComm:
    """torch.distributed communicator.  backend nccl (= RCCL over xGMI): device
    tensors go straight to the collective; gloo: staged through the host (CPU
    tests, or several ranks sharing one GPU in tests)."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.host_staging = dist.get_backend(group) == "gloo"

    @property
    def tapeable(self):
        """collectives issued through call() alone (here: none at all) - _lib.Tape
        may replay the code around them"""
        return self.size == 1

    def _h(self, t):
        return t.cpu() if (self.host_staging and t.is_cuda) else t

    def all_reduce_sum_(self, t):
        if self.size == 1:
            return t
        if self.host_staging and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)
        return t

    def all_reduce_begin(self, t, side_stream=None):
        """start an all-reduce whose result is needed only after all_reduce_end();
        torch.distributed: nothing to overlap with, the reduction happens here"""
        return self.all_reduce_sum_(t)

    def all_reduce_end(self):
        pass

    def all_to_all_counts(self, send_counts):
        """exchange one integer with every peer (setup only)"""
        if self.size == 1:
            return list(send_counts)
        dev = torch.device("cpu") if self.host_staging else \
            torch.device("cuda", torch.cuda.current_device())
        snd = torch.tensor(send_counts, dtype=torch.int64, device=dev)
        rcv = torch.empty(self.size, dtype=torch.int64, device=dev)
        dist.all_to_all_single(rcv, snd, group=self.group)
        return [int(v) for v in rcv.cpu()]

    def all_to_all_v(self, recv, send, recv_counts, send_counts, async_op=False):
        """recv/send: 1-D tensors; counts per peer.  Returns a waitable or None."""
        if self.size == 1:
            if recv.numel():
                recv.copy_(send[:recv.numel()])
            return None
        if self.host_staging and send.is_cuda:
            hs = send.cpu()
            hr = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(hr, hs, list(recv_counts), list(send_counts), group=self.group)
            recv.copy_(hr)
            return None
        return dist.all_to_all_single(recv, send, list(recv_counts), list(send_counts),
                                      group=self.group, async_op=async_op)


class RcclComm(TorchComm):
    """The data path straight on RCCL through the C ABI (gkoc_comm_*,
    csrc/comm.hip): an all-reduce or a halo exchange is one enqueue on the
    executor's stream - no process-group bookkeeping (20-45 us of host time per
    call through torch.distributed, tools/dist_host_cost.py) and no cross-stream
    event hops on the device.  torch.distributed (whatever backend is up) is used
    only to hand rank 0's communicator id to the other ranks and for the integer
    set-up exchanges.  Device buffers only."""

    direct = True

    def __init__(self, exec_, group=None):
        super().__init__(group)
        self.exec = exec_
        import os
        dev = torch.device("cpu") if self.host_staging else exec_.device

        def agree(ok, what):
            """every rank learns whether the step worked everywhere, so that a local
            failure raises on ALL ranks instead of leaving the others in a collective"""
            if self.size > 1:
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                ok = bool(flag.item())
            if not ok:
                raise GkoError(f"RcclComm: {what} failed on at least one rank: "
                               + lib().gkoc_last_error().decode(errors="replace"))

        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path = C.c_char_p(path.encode()) if os.path.exists(path) else None
        agree(lib().gkoc_comm_load_rccl(path) == 0 and not _inject_failure("load", self.rank),
              "binding librccl")
        ident = (C.c_uint8 * 128)()
        agree(self.rank != 0 or lib().gkoc_comm_unique_id(ident) == 0, "ncclGetUniqueId")
        if self.size > 1:
            t = torch.tensor(list(ident), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0,
                           group=group)
            ident = (C.c_uint8 * 128)(*t.cpu().tolist())
        self._handle = C.c_void_p(0)
        # every rank enters ncclCommInitRank (it is collective); its outcome is agreed on after
        rc = lib().gkoc_comm_create(C.byref(self._handle), C.c_int(self.size), C.c_int(self.rank), ident)
        try:
            agree(rc == 0 and not _inject_failure("init", self.rank), "ncclCommInitRank")
        except GkoError:
            self.close()
            raise
        self._cnt = {}

    tapeable = True

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            call("gkoc_comm_destroy", self._handle)
            self._handle = C.c_void_p(0)

    def all_reduce_sum_(self, t):
        if self.size == 1:
            return t
        if not (t.is_cuda and t.dtype.is_floating_point):
            return super().all_reduce_sum_(t)      # set-up integers
        call("gkoc_comm_all_reduce_sum", self._handle, self.exec.stream, t, t.numel(),
             C.c_size_t(t.element_size()))
        return t

    def all_reduce_begin(self, t, side_stream=None):
        """the all-reduce travels on side_stream; kernels enqueued on the executor's
        stream before all_reduce_end() overlap it (gkoc_comm_all_reduce_begin)"""
        if self.size == 1:
            return t
        side = C.c_void_p(side_stream.cuda_stream) if side_stream is not None else None
        call("gkoc_comm_all_reduce_begin", self._handle, self.exec.stream, side, t, t.numel(),
             C.c_size_t(t.element_size()))
        return t

    def all_reduce_end(self):
        if self.size > 1:
            call("gkoc_comm_all_reduce_end", self._handle, self.exec.stream)

    def _counts(self, counts):
        key = tuple(counts)
        arr = self._cnt.get(key)
        if arr is None:
            arr = self._cnt[key] = (C.c_int64 * len(key))(*key)
        return arr

    def exchange_begin(self, recv, send, recv_counts, send_counts, side_stream=None,
                       send_displs=None):
        """halo exchange of float buffers; kernels enqueued on the executor's stream
        before exchange_end overlap the transfer when side_stream is given.
        send_displs: per-peer offsets into `send` (then `send` is the vector itself
        and no pack kernel ran); None: `send` is packed in rank order."""
        side = C.c_void_p(side_stream.cuda_stream) if side_stream is not None else None
        call("gkoc_comm_exchange_begin", self._handle, self.exec.stream, side, send,
             self._counts(send_counts),
             self._counts(send_displs) if send_displs is not None else None, recv,
             self._counts(recv_counts), C.c_size_t(send.element_size()))

    def all_reduce_exchange_begin(self, t, recv, send, recv_counts, send_counts, side_stream,
                                  send_displs=None):
        """all_reduce_begin(t) and exchange_begin(...) behind ONE fork / join pair: the side stream
        reduces t, then exchanges; exchange_end / exchange_join end both (no all_reduce_end)"""
        call("gkoc_comm_all_reduce_exchange_begin", self._handle, self.exec.stream,
             C.c_void_p(side_stream.cuda_stream), t, t.numel(), C.c_size_t(t.element_size()), send,
             self._counts(send_counts), self._counts(send_displs) if send_displs is not None else None,
             recv, self._counts(recv_counts), C.c_size_t(send.element_size()))

    def fork_deferred(self, side_stream):
        """the fork of the NEXT begin call is opened by the caller's next kernel on the executor's
        stream (gkoc_comm_fork_deferred): returns the (word, number) objects to hand to that kernel
        (spmv_gated: fork=...); the word is NULL where the library prefers to fork by itself"""
        if not hasattr(self, "_fork_tok"):
            self._fork_tok = (C.c_void_p(0), C.c_uint32(0))
        w, n = self._fork_tok
        call("gkoc_comm_fork_deferred", self._handle, self.exec.stream, C.c_void_p(side_stream.cuda_stream),
             C.byref(w), C.byref(n))
        return self._fork_tok

    def topology(self):
        """gkoc_comm_topology_get as a dict: did the communicator see N ranks, on which devices, which RCCL"""
        t = CommTopology()
        call("gkoc_comm_topology_get", self._handle, C.byref(t))
        n = min(t.n_ranks, 16)
        return {"transport": "mailboxes" if t.transport == 1 else "rccl", "ranks": t.n_ranks,
                "ranks_seen": t.ranks_seen, "rccl_version": t.rccl_version or None,
                "cross_device": bool(t.cross_device),
                "window_uncached": bool(t.window_uncached) if t.transport == 1 else None,
                "gate_fence": t.gate_fence,
                "bus_ids": [t.bus_id[p].raw.split(b"\0", 1)[0].decode(errors="replace") for p in range(n)]}

    def check(self):
        """raises if a fork's poller ever gave up (gkoc_comm_fork_timed_out); synchronises"""
        flag = C.c_int(0)
        call("gkoc_comm_fork_timed_out", self._handle, C.byref(flag))
        if flag.value:
            raise GkoError("RcclComm: the exchange's stream waited a minute for the kernel that opens its "
                           "fork and went on without it (GKO_DEFERRED_FORK=0 forks in front of the product)")

    def exchange_end(self):
        call("gkoc_comm_exchange_end", self._handle, self.exec.stream)

    def exchange_join(self):
        """exchange_end that also waits for the kernels enqueued on the side stream behind the halo"""
        call("gkoc_comm_exchange_join", self._handle, self.exec.stream)

    def exchange_forget(self):
        """ends an exchange without a wait on the executor's stream: the reader of the halo waits
        for it itself (HipBackend.gate_open / spmv_gated)"""
        call("gkoc_comm_exchange_forget", self._handle)

    def all_to_all_v(self, recv, send, recv_counts, send_counts, async_op=False):
        if self.size == 1 or not send.is_cuda or not send.dtype.is_floating_point:
            return super().all_to_all_v(recv, send, recv_counts, send_counts, async_op)
        self.exchange_begin(recv, send, recv_counts, send_counts)
        return None


class IpcComm(RcclComm):
    """The same device-resident data path on the library's OWN transport (csrc/comm_ipc.hpp):
    mailboxes in peer-mapped device memory instead of RCCL.  An all-reduce is one kernel - every
    rank stores its values into every peer's window and sums what arrived in its own in rank order
    (same bits on every rank, every run); a halo exchange is one kernel that copies the boundary
    planes into the neighbours' windows and the neighbours' planes out of its own.  Stands where
    the reference lets a collective_communicator be chosen
    (include/ginkgo/core/distributed/collective_communicator.hpp:31-71).  Works between
    processes that SHARE one GPU (RCCL does not), so the whole N > 1 device path - forks, side
    stream, gated one-kernel product, pipelined solver steps - runs on a one-GPU box.
    torch.distributed only carries the window cards (handle + PCI bus id + kind of memory, all_gather) at
    set-up; gkoc_comm_ipc_connect refuses - on every rank alike - windows in plain device memory between
    DIFFERENT devices (default_comm then takes RCCL)."""

    def __init__(self, exec_, group=None, slot_bytes=0):
        TorchComm.__init__(self, group)
        self.exec = exec_
        dev = torch.device("cpu") if self.host_staging else exec_.device

        def agree(ok, what):
            if self.size > 1:
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                ok = bool(flag.item())
            if not ok:
                raise GkoError(f"IpcComm: {what} failed on at least one rank: "
                               + lib().gkoc_last_error().decode(errors="replace"))

        self._handle = C.c_void_p(0)
        mine = (C.c_uint8 * IPC_HANDLE_BYTES)()
        rc = lib().gkoc_comm_ipc_create(C.byref(self._handle), C.c_int(self.size), C.c_int(self.rank),
                                        C.c_int64(int(slot_bytes)), mine)
        try:
            agree(rc == 0 and not _inject_failure("load", self.rank), "creating / exporting the window")
            t = torch.tensor(list(mine), dtype=torch.uint8, device=dev)
            if self.size > 1:
                outs = [torch.empty_like(t) for _ in range(self.size)]
                dist.all_gather(outs, t, group=group)
            else:
                outs = [t]
            everybody = (C.c_uint8 * (IPC_HANDLE_BYTES * self.size))(
                *[int(v) for o in outs for v in o.cpu().tolist()])
            rc = lib().gkoc_comm_ipc_connect(self._handle, everybody)
            agree(rc == 0 and not _inject_failure("init", self.rank), "mapping the peers' windows")
        except GkoError:
            self.close(barrier=False)
            raise
        self._cnt = {}
        tr, unc = C.c_int(0), C.c_int(0)
        call("gkoc_comm_transport", self._handle, C.byref(tr), C.byref(unc))
        self.window_uncached = bool(unc.value)

    def close(self, barrier=True):
        """collective: no rank unmaps its window while a peer may still write into it"""
        if getattr(self, "_handle", None) is not None and self._handle.value:
            if barrier and self.size > 1 and dist.is_initialized():
                torch.cuda.synchronize(self.exec.device)
                try:
                    dist.barrier(group=self.group)
                except Exception:          # noqa: BLE001 - the group may be gone at interpreter exit
                    pass
            call("gkoc_comm_destroy", self._handle)
            self._handle = C.c_void_p(0)

    def set_patience_ms(self, ms):
        """patience of the waits enqueued from now on (0: GKOC_IPC_PATIENCE_MS / the default)"""
        call("gkoc_comm_set_patience_ms", self._handle, C.c_int64(int(ms)))

    def status(self):
        """0, or the bits of the waits that ran out of patience (gkoc_comm_status)"""
        st = C.c_uint32(0)
        call("gkoc_comm_status", self._handle, C.byref(st))
        return int(st.value)

    def check(self):
        super().check()
        self.exec.synchronize()
        st = self.status()
        if st:
            raise GkoError(f"IpcComm: a kernel of the mailbox transport stopped waiting for a peer (status {st:#x}: "
                           "1 all-reduce, 2 message, 4 acknowledgement) - the results since are not to be trusted")


def _inject_failure(stage, rank):
    """GKO_COMM_INJECT_FAIL="<rank>:<stage>" (tests only): make `stage` of the communicator
    bring-up fail on `rank` so that the all-ranks-together fallback can be exercised"""
    import os
    spec = os.environ.get("GKO_COMM_INJECT_FAIL", "")
    if not spec:
        return False
    r, _, st = spec.partition(":")
    return st == stage and r.isdigit() and int(r) == rank


def _try_comm(cls, exec_, base, group):
    """bring `cls` up and put a known-answer all-reduce through it; every decision is taken by ALL
    ranks together (a MIN all-reduce of "it worked here"); returns the communicator or None"""
    import os
    import sys
    ok, comm, probing = True, None, False
    try:
        comm = cls(exec_, group)
        probing = hasattr(comm, "set_patience_ms") and not os.environ.get("GKOC_IPC_PATIENCE_MS")
        if probing:
            # every rank is here (the hand-shake has just ended): a transport whose stores do not reach
            # the peers shows within seconds, not after the two minutes a late peer of a real job is given
            comm.set_patience_ms(20000)
        t = torch.full((2,), float(comm.rank + 1), dtype=torch.float64, device=exec_.device)
        comm.all_reduce_sum_(t)
        want_v = comm.size * (comm.size + 1) / 2
        ok = bool((t == want_v).all().item()) and not _inject_failure("answer", base.rank)
        if ok and hasattr(comm, "status"):
            ok = comm.status() == 0
        if probing:
            comm.set_patience_ms(0)
    except Exception as e:        # noqa: BLE001 - any failure means "not this one"
        print(f"[ginkgo_amd] rank {base.rank}: {cls.__name__} unavailable: {e}", file=sys.stderr)
        ok = False
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32,
                        device=torch.device("cpu") if base.host_staging else exec_.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if bool(flag.item()):
        return comm
    if ok:
        print(f"[ginkgo_amd] rank {base.rank}: {cls.__name__} is fine here but failed on another rank",
              file=sys.stderr)
    if comm is not None:
        try:
            comm.close(barrier=False) if isinstance(comm, IpcComm) else comm.close()
        except Exception:          # noqa: BLE001
            pass
    return None


def _iteration_cost_us(exec_, comm, n_elems, reps=30):
    """what one Cg iteration asks of the communicator - two 2-value all-reduces and one neighbour
    exchange of n_elems values - in microseconds, MAX over the ranks"""
    import time
    rank, size, dev = comm.rank, comm.size, exec_.device
    side = torch.cuda.Stream(device=dev)
    t = torch.ones(2, dtype=torch.float64, device=dev)
    peers = [p for p in (rank - 1, rank + 1) if 0 <= p < size]
    counts = [n_elems if p in peers else 0 for p in range(size)]
    send = torch.ones(n_elems * len(peers), dtype=torch.float64, device=dev)
    recv = torch.zeros(n_elems * len(peers), dtype=torch.float64, device=dev)
    for k in range(reps + 3):
        if k == 3:
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
        comm.all_reduce_sum_(t)
        comm.exchange_begin(recv, send, counts, counts, side)
        comm.exchange_end()
        comm.all_reduce_sum_(t)
        t.fill_(1.0)
    torch.cuda.synchronize(dev)
    us = torch.tensor([(time.perf_counter() - t0) / reps * 1e6], dtype=torch.float64,
                      device=torch.device("cpu") if comm.host_staging else dev)
    dist.all_reduce(us, op=dist.ReduceOp.MAX, group=comm.group)
    return float(us.item())


def default_comm(exec_, group=None, halo_elems=65536):
    """The communicator of the product path.  Device-resident candidates: IpcComm (the library's
    mailboxes in peer-mapped memory) and RcclComm (RCCL over xGMI through the C ABI).  Each one
    that comes up and passes a known-answer all-reduce on EVERY rank is timed on what a Cg
    iteration asks of it (two 2-value all-reduces + one halo exchange of `halo_elems` values, max
    over ranks) and the faster one is taken; none (gloo, one rank, GKO_COMM=torch, or failures -
    reported on stderr): torch.distributed itself.  Every decision is taken by all ranks together.
    GKO_COMM = ipc | rccl | torch forces one; with a gloo process group (ranks sharing one GPU in
    tests) only a forced ipc / rccl is tried.  The choice and the timings are kept in
    `default_comm.last` for the bench line."""
    import os
    import sys
    base = TorchComm(group)
    want = os.environ.get("GKO_COMM", "")
    default_comm.last = {"chosen": "TorchComm", "why": "one rank" if base.size == 1 else f"GKO_COMM={want or 'auto'}"}
    if base.size == 1 or want == "torch" or (base.host_staging and want not in ("rccl", "ipc")):
        return base
    classes = {"ipc": [IpcComm], "rccl": [RcclComm]}.get(want, [IpcComm, RcclComm])
    up = [c for c in (_try_comm(cls, exec_, base, group) for cls in classes) if c is not None]
    if not up:
        default_comm.last = {"chosen": "TorchComm", "why": "no device-resident communicator came up"}
        return base
    if len(up) == 1:
        default_comm.last = {"chosen": type(up[0]).__name__, "why": "the only one that came up" if not want else f"GKO_COMM={want}"}
        return up[0]
    costs = {}
    for c in up:
        try:
            with _Watchdog(120.0, f"timing {type(c).__name__}"):
                costs[type(c).__name__] = round(_iteration_cost_us(exec_, c, halo_elems), 1)
        except Exception as e:        # noqa: BLE001
            print(f"[ginkgo_amd] rank {base.rank}: timing {type(c).__name__} failed: {e}", file=sys.stderr)
            costs[type(c).__name__] = float("inf")
    best = min(up, key=lambda c: costs[type(c).__name__])      # the costs are max-over-ranks: same choice everywhere
    for c in up:
        if c is not best:
            try:
                c.close()
            except Exception:          # noqa: BLE001
                pass
    default_comm.last = {"chosen": type(best).__name__, "why": "faster on one Cg iteration's communication",
                         "iteration_comm_us": costs, "halo_elems": halo_elems}
    return best


default_comm.last = {}


class _Watchdog:
    """Abort the process (exit code 86) with a message when the guarded block does not finish
    in `seconds`: a hung collective cannot be cancelled, but the launcher can be told WHY the
    job died instead of being left to its own timeout."""

    on_fire = None      # bench.py: leave a line that says what was running before the process ends

    def __init__(self, seconds, what):
        self.seconds, self.what = seconds, what

    def __enter__(self):
        import os
        import sys
        import threading

        def fire():
            print(f"[ginkgo_amd] FATAL: {self.what} did not finish within {self.seconds} s "
                  f"(rank {os.environ.get('RANK', '0')}): a collective hangs - check that every rank "
                  "reached it, HSA_ENABLE_IPC_MODE_LEGACY=0, and the RCCL transport; "
                  "GKO_COMM=torch selects torch.distributed for the data path", file=sys.stderr)
            sys.stderr.flush()
            try:
                if _Watchdog.on_fire is not None:
                    _Watchdog.on_fire(self.what)
            except Exception:          # noqa: BLE001
                pass
            os._exit(86)

        self.t = threading.Timer(self.seconds, fire)
        self.t.daemon = True
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def comm_self_check(exec_, comm, n_elems=65536, reps=20, timeout_s=180.0, dtype=torch.float64):
    """Known-answer test of every collective form the solvers use, for ANY rank count, plus
    their latencies on this machine: (1) the 2-value all-reduce of DistributedCg, (2) the
    overlapped 3-value all-reduce of DistributedPipeCg (side stream, begin / end), (3) a
    neighbour exchange of `n_elems` values with rank - 1 and rank + 1 - the halo pattern of a
    slab partition - overlapped on the side stream when the communicator can.  Wrong data
    raises GkoError on all ranks together; a hang ends the process through _Watchdog.
    Returns {"all_reduce_us", "all_reduce_overlapped_us", "exchange_us", "communicator"}."""
    import time
    rank, size = comm.rank, comm.size
    dev = exec_.device
    side = torch.cuda.Stream(device=dev)
    direct = getattr(comm, "direct", False)
    res = {"communicator": type(comm).__name__, "ranks": size}
    bad = []
    with _Watchdog(timeout_s, "the communicator self-check"):
        # (1) in-stream all-reduce
        t = torch.empty(2, dtype=dtype, device=dev)
        for k in range(reps + 1):
            if k == 1:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            t[0] = rank + 1
            t[1] = 0.5 * (rank + 1)
            comm.all_reduce_sum_(t)
        torch.cuda.synchronize(dev)
        res["all_reduce_us"] = round((time.perf_counter() - t0) / reps * 1e6, 1)
        tot = size * (size + 1) / 2
        if t.cpu().tolist() != [tot, 0.5 * tot]:
            bad.append(f"all-reduce gave {t.cpu().tolist()}, expected {[tot, 0.5 * tot]}")
        # (2) overlapped all-reduce (PipeCg)
        t3 = torch.empty(3, dtype=dtype, device=dev)
        for k in range(reps + 1):
            if k == 1:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            t3.fill_(rank + 1)
            comm.all_reduce_begin(t3, side)
            comm.all_reduce_end()
        torch.cuda.synchronize(dev)
        res["all_reduce_overlapped_us"] = round((time.perf_counter() - t0) / reps * 1e6, 1)
        if t3.cpu().tolist() != [tot] * 3:
            bad.append(f"overlapped all-reduce gave {t3.cpu().tolist()}, expected {[tot] * 3}")
        # (3) neighbour exchange: the boundary planes of a slab partition
        peers = [p for p in (rank - 1, rank + 1) if 0 <= p < size]
        counts = [n_elems if p in peers else 0 for p in range(size)]
        send = torch.empty(n_elems * len(peers), dtype=dtype, device=dev)
        recv = torch.zeros(n_elems * len(peers), dtype=dtype, device=dev)
        for i, p in enumerate(peers):
            send[i * n_elems:(i + 1) * n_elems] = rank * 1000.0 + p     # "from rank to p"
        for k in range(reps + 1):
            if k == 1:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            if direct:
                comm.exchange_begin(recv, send, counts, counts, side)
                comm.exchange_end()
            else:
                comm.all_to_all_v(recv, send, counts, counts)
        torch.cuda.synchronize(dev)
        res["exchange_us"] = round((time.perf_counter() - t0) / reps * 1e6, 1)
        res["exchange_bytes_per_peer"] = n_elems * send.element_size()
        got = recv.cpu()
        for i, p in enumerate(peers):
            seg = got[i * n_elems:(i + 1) * n_elems]
            if not bool((seg == p * 1000.0 + rank).all()):
                bad.append(f"exchange: data from rank {p} is wrong (first value {float(seg[0])})")
        # everybody learns whether anybody failed
        flag = torch.tensor([0 if bad else 1], dtype=torch.int32,
                            device=torch.device("cpu") if comm.host_staging else dev)
        if size > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=comm.group)
    if not bool(flag.item()):
        raise GkoError(f"communicator self-check failed (rank {rank}: "
                       f"{'; '.join(bad) if bad else 'fine here, wrong on another rank'})")
    return res


class HipBackend:
    """numerical kernels of the distributed path on the Cdna4Executor"""

    tapeable = True        # every kernel method below is a plain call()

    def __init__(self, exec_):
        self.exec = exec_

    # -- storage
    def empty(self, n, dtype):
        return self.exec.alloc((n,), dtype)

    def zeros(self, n, dtype):
        return self.exec.zeros((n,), dtype)

    def index_tensor(self, array, dtype):
        return self.exec.to_device(np.asarray(array)).to(dtype)

    def vector(self, n, dtype=torch.float64):
        # zero-filled: callers use it as an initial guess (a recycled device block is
        # not zero)
        return Dense.create(self.exec, (n, 1), dtype).fill(0.0)

    def vector_from(self, array):
        return Dense.from_numpy(self.exec, array)

    def scalar(self, v, dtype=torch.float64):
        return scalar(self.exec, v, dtype)

    # -- matrix set-up
    def split(self, a, col_lo, col_hi, n_global):
        """a: Csr with the owned rows and global columns (device)."""
        ex = self.exec
        it, vt = IT[a.col_idxs.dtype], VT[a.dtype]
        n = a.size[0]
        idt = a.col_idxs.dtype
        col_map = ex.alloc((n_global + 1,), idt)
        local_ptrs = ex.alloc((n + 1,), idt, MEM_INDICES)
        nl_full = ex.alloc((n + 1,), idt)
        cnt = [C.c_int64(0) for _ in range(4)]
        call("gkoc_dist_split_count_" + it, ex.stream, n, a.row_ptrs, a.col_idxs, col_lo,
             col_hi, n_global, col_map, local_ptrs, nl_full, *[C.byref(c) for c in cnt])
        n_halo, nnz_l, nnz_nl, n_nl_rows = (c.value for c in cnt)
        local_cols, local_vals = ex.alloc((nnz_l,), idt, MEM_INDICES), ex.alloc((nnz_l,), a.dtype, MEM_VALUES)
        nl_rows, nl_ptrs = ex.alloc((n_nl_rows,), idt), ex.alloc((n_nl_rows + 1,), idt)
        nl_cols, nl_vals = ex.alloc((nnz_nl,), idt), ex.alloc((nnz_nl,), a.dtype)
        recv_gidx = ex.alloc((n_halo,), idt)
        call(f"gkoc_dist_split_fill_{vt}_{it}", ex.stream, n, a.row_ptrs, a.col_idxs, a.values,
             col_lo, col_hi, n_global, col_map, local_ptrs, nl_full, local_cols, local_vals,
             nl_rows, nl_ptrs, nl_cols, nl_vals, recv_gidx)
        local = Csr(ex, (n, col_hi - col_lo), local_vals, local_cols, local_ptrs)
        # the arena has put values / indices / vectors into different memory classes
        # (DESIGN.md 3.2); kept for the bench line
        self.placement_log = local.memory_classes() if nnz_l > 0 else None
        nl = dict(rows=nl_rows, ptrs=nl_ptrs, cols=nl_cols, vals=nl_vals, n=n_nl_rows,
                  suffix=f"{vt}_{it}")
        # Contiguous partitions of stencil-like matrices: the rows with non-local entries are the
        # first k and the last m local rows.  They are kept a second time as COMPLETE rows (own
        # and halo columns in the original order), so that the local SpMV can leave them out and
        # they can be computed on the exchange's stream as soon as the halo is in
        # (DistributedMatrix.apply, gkoc_csr_rowlist_spmv_full_*).
        if n_nl_rows > 0:
            rows_h = nl_rows.cpu().numpy().astype(np.int64)
            k = int(np.searchsorted(rows_h, n // 2)) if n > 1 else n_nl_rows
            head, tail = rows_h[:k], rows_h[k:]
            if np.array_equal(head, np.arange(len(head))) and \
                    np.array_equal(tail, np.arange(n - len(tail), n)):
                f_ptrs = ex.alloc((n_nl_rows + 1,), idt)
                nnz_f = C.c_int64(0)
                call("gkoc_dist_boundary_count_" + it, ex.stream, n_nl_rows, nl_rows, a.row_ptrs, f_ptrs,
                     C.byref(nnz_f))
                f_cols, f_vals = ex.alloc((nnz_f.value,), idt), ex.alloc((nnz_f.value,), a.dtype)
                # halo entry h has column halo_base + h: the local size rounded up to 128 bytes, so that
                # a halo kept BEHIND the local vector (the one-kernel product below) starts on a cache
                # line of its own - the last local entries and the first halo entries never share one
                n_local = col_hi - col_lo
                halo_base = -(-n_local // 32) * 32
                call(f"gkoc_dist_boundary_fill_{vt}_{it}", ex.stream, n_nl_rows, nl_rows, a.row_ptrs,
                     a.col_idxs, a.values, col_lo, col_hi, halo_base, col_map, f_ptrs, f_cols, f_vals)
                nl["full"] = dict(ptrs=f_ptrs, cols=f_cols, vals=f_vals, n_local=n_local, halo_base=halo_base,
                                  interior=(len(head), n - len(tail)), head=len(head), tail=len(tail),
                                  n_halo=n_halo)
                # The product in one kernel (gkoc_csr_spmv_gated_*) reads the interior rows from the
                # local block and these complete boundary rows from the same launch: no further copy of
                # the matrix (round 3 kept ALL rows a second time).  It exists where the boundary rows
                # are few enough for their waves to wait on the device without starving the exchange.
                if os.environ.get("GKO_GATED_SPMV", "1") != "0" and nnz_l > 0:
                    nl["full"]["gated"] = bool(lib().gkoc_csr_spmv_gated_fits(
                        C.c_int64(n), C.c_int64(len(head)), C.c_int64(len(tail))))
        return local, nl, recv_gidx

    def to_host(self, t):
        return t.cpu().numpy()

    # -- apply pieces
    def gather(self, x, idx, out):
        if idx.numel():
            x.row_gather(idx, out)

    def spmv(self, a, x, y):
        a.apply(x, y)

    def rowlist_add(self, nl, halo, y):
        if nl["n"]:
            call("gkoc_csr_rowlist_spmv_add_" + nl["suffix"], self.exec.stream, nl["n"],
                 nl["rows"], nl["ptrs"], nl["cols"], nl["vals"], halo.values, halo.ld,
                 y.values, y.ld, y.size[1])

    def spmv_rows(self, a, r0, r1, x, y):
        """y[r0:r1] = A[r0:r1, :] x  (the interior rows of a slab: a CSR over the same value /
        column arrays whose row pointers start at row r0)"""
        if r1 <= r0:
            return
        key = ("rows", r0, r1, y.values.data_ptr())
        view = a.__dict__.setdefault("_row_views", {}).get(key)
        if view is None:
            sub = Csr(self.exec, (r1 - r0, a.size[1]), a.values, a.col_idxs, a.row_ptrs[r0:r1 + 1])
            view = a._row_views[key] = (sub, Dense(self.exec, y.values[r0:r1]))
        view[0].apply(x, view[1])

    def rowlist_full(self, nl, x, halo, y, stream=None):
        """y[boundary rows] = their complete row sums over [x | halo]; on `stream` (the
        exchange's stream) when given"""
        f = nl["full"]
        st = C.c_void_p(stream.cuda_stream) if stream is not None else self.exec.stream
        call("gkoc_csr_rowlist_spmv_full_" + nl["suffix"], st, nl["n"], nl["rows"], f["ptrs"],
             f["cols"], f["vals"], f["halo_base"], x.values, halo.values, y.values)

    def gate_new(self):
        """the word of gkoc_csr_spmv_gated_* / gkoc_gate_open (device) and the exchange count (host)"""
        return self.exec.zeros((2,), torch.int32), C.c_uint32(0)

    def gate_open(self, stream, gate):
        """on `stream` (the exchange's stream): the halo in front of this call has arrived; counts
        the exchange (also in replays of a recorded iteration)"""
        bump(gate[1])
        call("gkoc_gate_open", C.c_void_p(stream.cuda_stream), gate[0], gate[1])

    def spmv_gated(self, local, nl, x_ext, y, gate, fork=None):
        """(fork: the (word, number) of comm.fork_deferred - the kernel's first wave opens the
        exchange's fork)
        y = A [x | halo] for ALL local rows in one kernel: the interior rows from the local
        block, the boundary rows (complete rows, nl["full"]) on its last waves, which wait for the
        gate_open in front of this call; x_ext: the local vector with the halo behind it, starting
        at entry nl["full"]["halo_base"]"""
        f = nl["full"]
        call("gkoc_csr_spmv_gated_" + nl["suffix"], self.exec.stream, y.size[0], local.row_ptrs,
             local.col_idxs, local.values, f["ptrs"], f["cols"], f["vals"], x_ext, y.values, f["head"],
             f["tail"], gate[0], gate[1], *(fork or (None, C.c_uint32(0))))

    def spmv_gated_dot(self, local, nl, x_ext, y, gate, out, fork=None):
        """spmv_gated and out = LOCAL <x, y> from the waves that hold the row sums (one partial sum
        per wave, one fold launch): gkoc_x_csr_spmv_gated_dot_*"""
        f = nl["full"]
        w, wb = self._xwork(y.size[0] + 128, y.dtype)
        call("gkoc_x_csr_spmv_gated_dot_" + nl["suffix"], self.exec.stream, y.size[0], local.row_ptrs,
             local.col_idxs, local.values, f["ptrs"], f["cols"], f["vals"], x_ext, y.values, f["head"],
             f["tail"], gate[0], gate[1], *(fork or (None, C.c_uint32(0))), out.values, w, wb)

    def spmv_dot(self, a, x, y, out):
        """y = A_local x and out = local <x, y> in one pass; False if there is no such kernel
        for this operand layout"""
        if not (isinstance(a, Csr) and a.size[0] == a.size[1] and x.ld == 1 and y.ld == 1 and
                x.size[1] == 1):
            return False
        w, _ = self._xwork(x.size[0], x.dtype)
        a.apply_dot(x, y, out, w)
        return True

    def rowlist_add_dot(self, nl, halo, y, x, out):
        """y[rows] += A_nl halo and out += <x[rows], what was added>"""
        if not nl["n"]:
            return
        w = nl.get("dot_work")
        need = (nl["n"] + 63) // 64
        if w is None or w.numel() < need or w.dtype != y.dtype:
            w = nl["dot_work"] = self.exec.alloc((need,), y.dtype)
        call("gkoc_x_csr_rowlist_spmv_add_dot_" + nl["suffix"], self.exec.stream, nl["n"], nl["rows"],
             nl["ptrs"], nl["cols"], nl["vals"], halo.values, y.values, x.values, out.values, w,
             C.c_size_t(w.numel() * w.element_size()))

    def jacobi(self, a, max_block_size):
        return Jacobi.build().with_max_block_size(max_block_size).on(self.exec).generate(a)

    # -- Krylov pieces (thin wrappers so that tests can swap the backend)
    def cg_initialize(self, b, r, z, p, q, prev_rho, rho, stop):
        call("gkoc_cg_initialize_" + VT[b.dtype], self.exec.stream, b.size[0], 1, b.values, b.ld,
             r.values, r.ld, z.values, z.ld, p.values, p.ld, q.values, q.ld, prev_rho.values,
             rho.values, stop)

    def cg_step_1(self, p, z, rho, prev_rho, stop):
        call("gkoc_cg_step_1_" + VT[p.dtype], self.exec.stream, p.size[0], 1, p.values, p.ld,
             z.values, z.ld, rho.values, prev_rho.values, stop)

    def check_slot(self):
        """a token for cg_step_1_check / check_done: the next slot of the pinned flag ring"""
        return self._check_slot()

    def cg_step_1_check(self, p, z, rho, prev_rho, tau, tau0, factor, stop, slot, squared=True):
        """the criterion on tau (as check_begin; its answer arrives in `slot`) and cg::step_1 in
        ONE kernel (gkoc_x_cg_step_1_check_*); one column, unit strides"""
        call("gkoc_x_cg_step_1_check_" + VT[p.dtype], self.exec.stream, p.size[0], p.values, z.values,
             rho.values, prev_rho.values, tau.values, tau0.values,
             C.c_double(factor) if p.dtype == torch.float64 else C.c_float(factor),
             C.c_int(1 if squared else 0), C.c_uint8(2), C.c_int(1), stop, self._chk_host[slot])

    def cg_step_2(self, x, r, p, q, beta, rho, stop):
        call("gkoc_cg_step_2_" + VT[x.dtype], self.exec.stream, x.size[0], 1, x.values, x.ld,
             r.values, r.ld, p.values, p.ld, q.values, q.ld, beta.values, rho.values, stop)

    def local_dot(self, x, y, out):
        x.compute_dot(y, out)

    def pipe_cg_initialize_1(self, b, r, prev_rho, stop):
        call("gkoc_pipe_cg_initialize_1_" + VT[b.dtype], self.exec.stream, b.size[0], 1, b.values,
             b.ld, r.values, r.ld, prev_rho.values, stop)

    def pipe_cg_initialize_2(self, p, q, f, g, beta, z, w, m, n, delta):
        call("gkoc_pipe_cg_initialize_2_" + VT[p.dtype], self.exec.stream, p.size[0], 1, p.values,
             p.ld, q.values, q.ld, f.values, f.ld, g.values, g.ld, beta.values, z.values, z.ld,
             w.values, w.ld, m.values, m.ld, n.values, n.ld, delta.values)

    def pipe_cg_step_1(self, x, r, z, w, p, q, f, g, rho, beta, stop):
        call("gkoc_pipe_cg_step_1_" + VT[x.dtype], self.exec.stream, x.size[0], 1, x.values, x.ld,
             r.values, r.ld, z.values, z.ld, z.values, z.ld, w.values, w.ld, p.values, p.ld,
             q.values, q.ld, f.values, f.ld, g.values, g.ld, rho.values, beta.values, stop)

    def pipe_cg_step_1_dots(self, x, r, z, w, p, q, f, g, rho, beta, stop, out3):
        """pipe_cg::step_1 and out3 = local {<r,z>, <w,z>, <r,r>} in one pass; False if the
        layout has no fused kernel"""
        if not all(v.ld == 1 and v.size[1] == 1 for v in (x, r, z, w, p, q, f, g)):
            return False
        wk, wb = self._xwork(x.size[0], x.dtype)
        call("gkoc_x_pipe_cg_step_1_dots_" + VT[x.dtype], self.exec.stream, x.size[0], x.values,
             r.values, z.values, w.values, p.values, q.values, f.values, g.values, rho.values,
             beta.values, stop, out3, wk, wb)
        return True

    def step_gate(self, gate, tau, tau0, factor, stop, slot):
        """a recorded-call argument for the fused PipeCg step kernels: wait for `gate` (the pair of
        gate_new; its exchange number is read when the call is issued, also in replays) and evaluate
        ImplicitResidualNorm on tau (= ||r||^2) into flag slot `slot` (check_slot / check_done)"""
        sg = StepGate(gate[0].data_ptr() if gate is not None else None, 0, 1, tau.values.data_ptr(),
                      tau0.values.data_ptr(), float(factor), self._chk_host[slot].data_ptr(), 2, 1)
        keep = (gate, tau, tau0, stop)

        class _Arg:
            """passed by reference; wait_number is refreshed from the gate's counter on every use"""
            def __init__(self):
                self.sg, self.keep = sg, keep

            @property
            def _as_parameter_(self):
                if gate is not None:
                    sg.wait_number = gate[1].value
                return C.c_void_p(C.addressof(sg))
        return _Arg()

    def pipe_cg_step_2_step_1_dots(self, x, r, z, w, p, q, f, g, m, n, prev_rho, rho, delta, beta_in,
                                   beta_out, stop, out3, gate=None):
        """pipe_cg::step_2 of this iteration and step_1 of the next one in one pass (ten vectors in,
        eight out instead of 12 + 12), out3 = local {<r,z>, <w,z>, <r,r>} of the new vectors; False
        if the layout has no such kernel"""
        if not all(v.ld == 1 and v.size[1] == 1 for v in (x, r, z, w, p, q, f, g, m, n)):
            return False
        wk, wb = self._xwork(x.size[0], x.dtype)
        call("gkoc_x_pipe_cg_step_2_step_1_dots_" + VT[x.dtype], self.exec.stream, x.size[0], x.values,
             r.values, z.values, w.values, p.values, q.values, f.values, g.values, m.values, n.values,
             prev_rho.values, rho.values, delta.values, beta_in.values, beta_out.values, stop, out3, wk, wb,
             gate)
        return True

    def pipe_cg_steps_jacobi(self, m_op, x, r, z, w, p, q, f, g, m, n, prev_rho, rho, delta, beta_in,
                             beta_out, stop, out3, probe=False, gate=None):
        """the same AND m = M w (block-Jacobi) in one kernel: the new w goes from the registers
        that computed it into the block product; False if this preconditioner / layout has no
        such kernel (probe=True: only answer)"""
        if not (hasattr(m_op, "can_fuse_step_2") and m_op.can_fuse_step_2(r) and
                all(v.ld == 1 and v.size[1] == 1 for v in (x, r, z, w, p, q, f, g, m, n))):
            return False
        if probe:
            return True
        wk, wb = self._xwork(x.size[0], x.dtype)
        call("gkoc_x_pipe_cg_steps_jacobi_" + m_op._suf, self.exec.stream, m_op.num_blocks, m_op.size[0],
             C.c_uint32(m_op.max_block_size), m_op.scheme, m_op.block_pointers, m_op.blocks, x.values,
             r.values, z.values, w.values, p.values, q.values, f.values, g.values, m.values, n.values,
             prev_rho.values, rho.values, delta.values, beta_in.values, beta_out.values, stop, out3, wk, wb,
             gate)
        return True

    def pipe_cg_step_2(self, beta, p, q, f, g, z, w, m, n, prev_rho, rho, delta, stop):
        call("gkoc_pipe_cg_step_2_" + VT[p.dtype], self.exec.stream, p.size[0], 1, beta.values,
             p.values, p.ld, q.values, q.ld, f.values, f.ld, g.values, g.ld, z.values, z.ld,
             w.values, w.ld, m.values, m.ld, n.values, n.ld, prev_rho.values, rho.values,
             delta.values, stop)

    def scalar_tuple(self, k, dtype=torch.float64):
        """k adjacent device scalars: (k-element tensor, [view of [0], ..., view of [k-1]])"""
        t = self.exec.zeros((k,), dtype)
        return t, [Dense(self.exec, t[i:i + 1].view(1, 1)) for i in range(k)]

    # fused producer + local reduction (gkoc_x_*); vectors bit-identical
    def _xwork(self, n, dtype):
        w = getattr(self, "_xw", None)
        nbytes = lib().gkoc_x_workspace_bytes(C.c_int64(n), C.c_size_t(torch.empty((), dtype=dtype).element_size()))
        if w is None or w.numel() * w.element_size() < nbytes or w.dtype != dtype:
            es = torch.empty((), dtype=dtype).element_size()
            w = self._xw = self.exec.alloc(((nbytes + es - 1) // es,), dtype)
        return w, C.c_size_t(w.numel() * w.element_size())

    def jacobi_apply_dot(self, m, r, z, out):
        """z = M r, out = local <r, z>; False if this preconditioner layout has
        no fused kernel (caller uses apply + local_dot)"""
        if not (hasattr(m, "can_fuse_dot") and m.can_fuse_dot(r) and z.ld == 1):
            return False
        w, _ = self._xwork(r.size[0], r.dtype)
        m.apply_dot(r, z, out, w)
        return True

    def cg_step_2_sqnorm(self, x, r, p, q, beta, rho, stop, out):
        """cg::step_2 and out = local ||r_new||^2"""
        if not all(v.ld == 1 and v.size[1] == 1 for v in (x, r, p, q)):
            return False
        w, wb = self._xwork(x.size[0], x.dtype)
        call("gkoc_x_cg_step_2_norm_" + VT[x.dtype], self.exec.stream, x.size[0], x.values,
             r.values, p.values, q.values, beta.values, rho.values, stop, out.values,
             C.c_int(0), w, wb)
        return True

    def cg_step_2_jacobi(self, m, x, r, p, q, beta, rho, stop, z, rho_out, sq_out):
        """cg::step_2, z = M r_new, rho_out = local <r,z>, sq_out = local ||r||^2 in one kernel
        (gkoc_x_cg_step_2_jacobi_apply_*); False if this preconditioner layout has no such kernel"""
        if not (hasattr(m, "can_fuse_step_2") and m.can_fuse_step_2(r) and
                all(v.ld == 1 and v.size[1] == 1 for v in (x, r, p, q, z))):
            return False
        w, _ = self._xwork(x.size[0], x.dtype)
        m.step_2_apply_dot(x, r, p, q, beta, rho, stop, z, rho_out, sq_out, False, w)
        return True

    def local_sqnorm(self, x, out):
        x.compute_squared_norm2(out)

    def sqrt_(self, s):
        call("gkoc_dense_compute_sqrt_" + VT[s.dtype], self.exec.stream, 1, s.values)

    def stop_flags(self):
        return self.exec.zeros((2,), torch.uint8), self.exec.zeros((1,), torch.uint8)

    def scalar_pair(self, dtype=torch.float64):
        """two adjacent device scalars: (2-element tensor, view of [0], view of [1])"""
        t = self.exec.zeros((2,), dtype)
        return t, Dense(self.exec, t[0:1].view(1, 1)), Dense(self.exec, t[1:2].view(1, 1))

    # asynchronous criterion check: the kernel writes its two flags straight into
    # pinned host memory (one slot per in-flight check), an event marks it done;
    # the host looks at the answer when it wants to
    _NSLOT = 16

    def check_begin(self, tau, tau0, factor, stop, squared=False):
        """squared: tau holds ||r||^2; the criterion kernel of ImplicitResidualNorm
        (sqrt(|tau|) <= factor * tau0, residual_norm.cpp:209-230) then saves the
        separate sqrt launch.
        No event is recorded behind the kernel: on this hardware an event record is a barrier
        packet that leaves the device idle for ~6 us before the next kernel starts (kernel
        timelines of an 8-rank iteration, profiles/r03_dist_sim_timelines.txt).  The kernel writes
        its two flag bytes into a pinned (host-coherent) slot that the host has set to 0xFF; the
        host polls the slot when it wants the answer - `check_lag` iterations later, when the
        bytes have long arrived."""
        slot = self._check_slot()
        key = (tau.values.data_ptr(), tau0.values.data_ptr(), stop.data_ptr(), factor, slot, squared)
        tape = self._chk_tapes.get(key)
        if tape is None:
            if len(self._chk_tapes) > 4 * self._NSLOT:
                self._chk_tapes.clear()
            name = "gkoc_implicit_residual_norm_" if squared else "gkoc_residual_norm_"
            with _record() as tape:
                call(name + VT[tau.dtype], self.exec.stream, 1, tau.values, tau0.values,
                     C.c_double(factor) if tau.dtype == torch.float64 else C.c_float(factor),
                     C.c_uint8(2), C.c_int(1), stop, self._chk_host[slot], None, None)
            self._chk_tapes[key] = tape
        else:
            tape.replay()
        return slot

    def _check_slot(self):
        """the next slot of the pinned flag ring, marked 'no answer yet'"""
        if not hasattr(self, "_chk_host"):
            self._chk_host = torch.zeros((self._NSLOT, 2), dtype=torch.uint8).pin_memory()
            self._chk_np = self._chk_host.numpy()
            self._chk_next = 0
            self._chk_tapes = {}
        slot = self._chk_next
        self._chk_next = (slot + 1) % self._NSLOT
        self._chk_np[slot, 0] = self._chk_np[slot, 1] = 0xFF
        return slot

    def check_done(self, token, block=True):
        flags = self._chk_np[token]
        spins = 0
        while flags[0] == 0xFF or flags[1] == 0xFF:
            spins += 1
            if spins == 2000:
                # not there yet: wait for the stream instead of burning the core (the kernel's
                # stores are visible at the latest when the stream has drained)
                self.exec.synchronize()
            elif spins > 2000 and spins % 2000 == 0:
                raise GkoError("criterion flags did not arrive in pinned memory")
        return bool(flags[0])

    max_check_lag = 6       # must stay below _NSLOT
    check_takes_squared_norm = True

    def residual_check(self, tau, tau0, factor, stop, flags):
        allc, chg = C.c_int(0), C.c_int(0)
        call("gkoc_residual_norm_" + VT[tau.dtype], self.exec.stream, 1, tau.values, tau0.values,
             C.c_double(factor) if tau.dtype == torch.float64 else C.c_float(factor),
             C.c_uint8(2), C.c_int(1), stop, flags, C.byref(allc), C.byref(chg))
        return bool(allc.value)

    def synchronize(self):
        self.exec.synchronize()

    def side_stream(self):
        # high priority: what runs here (RCCL's send / recv and all-reduce kernels, the boundary
        # rows) are a few workgroups that must get onto the device WHILE the local SpMV fills it;
        # with equal priority they queue behind its tens of thousands of workgroups
        return torch.cuda.Stream(device=self.exec.device, priority=-1)


class DistributedMatrix:
    """distributed::Matrix for a contiguous row partition.

    `owned` holds this rank's rows (CSR, GLOBAL column indices).  Set-up follows
    read_distributed (matrix.cpp:300-381); apply follows apply_impl (:450-509)."""

    def __init__(self, backend, comm, partition, owned, local_format="csr"):
        """local_format: "csr", or "sellp" - the local block is converted to matrix::Sellp
        (slice size 64) and multiplied in that format (BASELINE configs[4]: SELL-P vs CSR); the
        non-local block stays a row list"""
        self.backend, self.comm, self.partition = backend, comm, partition
        self.rank = comm.rank
        lo, hi = partition.range_of(self.rank)
        if owned.size[0] != hi - lo:
            raise GkoError("DistributedMatrix: owned rows do not match the partition")
        self.n_local, self.n_global = hi - lo, partition.n_global
        self.dtype = owned.dtype
        self.local, self.nl, recv_gidx = backend.split(owned, lo, hi, self.n_global)
        # ---- exchange plan (RowGatherer ctor, row_gatherer.cpp:283-314)
        gidx = backend.to_host(recv_gidx).astype(np.int64)
        owners = partition.owner_of(gidx)
        self.recv_counts = [int(np.sum(owners == p)) for p in range(comm.size)]
        self.send_counts = comm.all_to_all_counts(self.recv_counts)
        idt = recv_gidx.dtype
        send_gidx = backend.empty(sum(self.send_counts), torch.int64)
        want = backend.index_tensor(gidx, torch.int64)
        comm.all_to_all_v(send_gidx, want, self.send_counts, self.recv_counts)
        self.send_idx = (send_gidx - lo).to(idt)
        # peers that want one contiguous row range each (slab partitions): the
        # vector itself can be the send buffer (send_offsets of i_all_to_all_v)
        sidx = backend.to_host(self.send_idx).astype(np.int64)
        self.send_displs, pos = [], 0
        for c in self.send_counts:
            seg = sidx[pos:pos + c]
            if c and not np.array_equal(seg, np.arange(seg[0], seg[0] + c)):
                self.send_displs = None
                break
            self.send_displs.append(int(seg[0]) if c else 0)
            pos += c
        self.n_halo, self.n_send = len(gidx), sum(self.send_counts)
        self.send_buf = backend.vector(self.n_send, self.dtype)
        self.recv_buf = backend.vector(self.n_halo, self.dtype)
        self._side = backend.side_stream() if hasattr(backend, "side_stream") else None
        self.global_nnz = None
        self.fused_dot_min_rows = 1 << 22
        import os
        # boundary rows as complete rows over [x | halo] (single-domain bits, overlapped with the
        # local SpMV) or, GKO_FULL_BOUNDARY=0, the round-2 form: whole local block, then
        # boundary rows += halo part
        self.use_full_boundary = os.environ.get("GKO_FULL_BOUNDARY", "1") != "0"
        self.local_format = local_format
        self.local_op = self.local
        if local_format == "sellp":
            self.local_op = self.local.convert_to_sellp()
            self.use_full_boundary = False       # (the row-range products are CSR kernels)
        elif local_format != "csr":
            raise GkoError(f"DistributedMatrix: unknown local format {local_format}")
        # <p, q> from the product's waves (gkoc_x_csr_spmv_gated_dot_*): measured per rank of 8 on
        # 256^3, it costs the product 8 us and its fold 4 - the separate dot + fold cost 9.  Off.
        self.gated_dot = os.environ.get("GKO_GATED_DOT", "0") != "0"
        self.deferred_fork = os.environ.get("GKO_DEFERRED_FORK", "1") != "0"
        # the whole product in ONE kernel whose last waves (the boundary rows) wait for the halo by
        # themselves: for vectors from ext_vector() (the solvers' search directions), a
        # device-resident communicator and zero-copy send planes
        self._gate = (backend.gate_new() if self.nl.get("full", {}).get("gated") and hasattr(backend, "spmv_gated") and
                      self._side is not None and getattr(comm, "direct", False) and
                      hasattr(comm, "exchange_forget") and self.send_displs is not None and
                      self.use_full_boundary else None)

    def ext_vector(self):
        """a zero local vector whose storage continues with room for the halo: apply() of such a
        vector receives the halo right behind it and runs as one kernel (gkoc_csr_spmv_gated_*).
        An ordinary vector where that path does not exist."""
        be = self.backend
        if self._gate is None:
            return be.vector(self.n_local, self.dtype)
        base = self.nl["full"]["halo_base"]          # the halo starts on a 128-byte boundary
        store = be.exec.zeros((base + max(self.n_halo, 1),), self.dtype)
        v = Dense(be.exec, store[:self.n_local].view(self.n_local, 1))
        v._ext_store, v._ext_halo = store, store[base:base + self.n_halo]
        return v

    def check_gate(self):
        """raises if a boundary wave of the one-kernel product ever gave up waiting for its halo
        (synchronises; the solvers call it when they return)"""
        if self._gate is not None and int(self._gate[0][1].item()) != 0:
            raise GkoError("DistributedMatrix: the halo exchange did not arrive within the one-kernel "
                           "product's patience (GKO_GATED_SPMV=0 selects the join-based product)")
        if self._gate is not None and hasattr(self.comm, "check"):
            self.comm.check()

    def conservative(self):
        """from now on: the join-based product (local rows || exchange, boundary rows in stream order,
        forks in front of the exchange) - the reference's shape, nothing waits inside a kernel"""
        self._gate = None
        self.deferred_fork = False

    def self_check(self, seed=7, rounds=1):
        """The one-kernel product against the join-based one on THIS communicator, before anything
        is timed: same bits on every local row, no boundary wave that gave up, no fork that timed out.
        Every rank runs the same collectives.  (ok, what) - on failure the caller agrees with the other
        ranks and calls conservative() on all of them (agreed_self_check does both).  rounds > 1: a
        short soak - a NEW x (so a new halo from every neighbour) per round, each compared."""
        if self._gate is None or self.comm.size == 1:
            return True, "join-based product (no one-kernel product for this matrix / communicator)"
        be = self.backend
        x = self.ext_vector()
        y1, y2 = be.vector(self.n_local, self.dtype), be.vector(self.n_local, self.dtype)
        rng = np.random.default_rng(seed + self.rank)
        try:
            for r in range(max(1, int(rounds))):
                x.values.copy_(torch.from_numpy(rng.uniform(-1, 1, self.n_local)).view(-1, 1))
                for _ in range(3 if r == 0 else 1):
                    self.apply(x, y1)
                self.check_gate()
                gate, self._gate = self._gate, None
                try:
                    self.apply(x, y2)
                finally:
                    self._gate = gate
                be.synchronize()
                if not torch.equal(y1.values, y2.values):
                    return False, f"one-kernel product differs from the join-based product (round {r})"
        except GkoError as e:
            return False, str(e)[:200]
        return True, ("one-kernel product == join-based product bit for bit" +
                      (f" in {rounds} rounds with fresh halos" if rounds > 1 else ""))

    def agreed_self_check(self, soak_rounds=64):
        """self_check + the agreement of all ranks + its consequences, collectively: a rank that sees a
        difference (or a wave that gave up, a fork that timed out) sends ALL ranks to the join-based product.
        With a peer on ANOTHER device (comm.topology()) the gated kernels pay a system-scope acquire per
        waiting wave (csrc/common.hpp gate_fence_policy) and the check is a soak of `soak_rounds` products with
        fresh halos; only when that has passed on every rank is the cheap gate trusted on this communicator
        (GKO_GATE_TRUST=0 keeps the fence whatever the soak says).  Returns what the bench line prints."""
        import os
        topo = self.comm.topology() if hasattr(self.comm, "topology") else None
        cross = bool(topo and topo["cross_device"])
        ok, why = self.self_check(rounds=soak_rounds if cross and self._gate is not None else 1)
        if self.comm.size > 1:
            flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64,
                                device=torch.device("cpu") if self.comm.host_staging else self.backend.exec.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.comm.group)
            all_ok = float(flag.item()) >= 1.0
        else:
            all_ok = ok
        if not all_ok:
            self.conservative()
        fence = None
        if topo is not None:
            if all_ok and cross and self._gate is not None and os.environ.get("GKO_GATE_TRUST", "1") != "0":
                fence = gate_fence_policy(0)
            else:
                fence = gate_fence_policy(-1)
        return {"one_kernel_product": self._gate is not None,
                "self_check": why if all_ok or not ok else "failed on another rank",
                "peers_on_other_devices": cross,
                "gate_fence": {None: None, 0: "cheap gate (acquire only for a wave that waited)" +
                               (", trusted after the soak on this communicator" if cross else ""),
                               1: "agent-scope acquire per waiting wave",
                               2: "system-scope acquire per waiting wave (a peer on another device)"}[fence]}

    def _fork_token(self):
        """the exchange about to begin is forked by the product's own kernel where the
        communicator offers it (no event, no kernel in front of the product on the main queue)"""
        if self.deferred_fork and hasattr(self.comm, "fork_deferred"):
            return self.comm.fork_deferred(self._side)
        return None

    def _gated(self, x, y):
        return (self._gate is not None and getattr(x, "_ext_halo", None) is not None and
                x.ld == 1 and y.ld == 1 and x.size[1] == 1 and self.use_full_boundary)

    def apply_dot(self, x, y, out):
        """y_local = A[owned rows, :] x and out = LOCAL part of <x, y> (the caller all-reduces):
        the local block through the fused SpMV + dot, the boundary rows' share next to their
        update.  False if the backend has no such kernels (then nothing was done)."""
        be = self.backend
        # the one-kernel product leaves one partial sum per wave; ONE launch folds them
        if (self.comm.size > 1 and self._gated(x, y) and hasattr(be, "spmv_gated_dot") and
                self.gated_dot):
            fork = self._fork_token()
            self.comm.exchange_begin(x._ext_halo, x.values, self.recv_counts, self.send_counts, self._side,
                                     self.send_displs)
            be.gate_open(self._side, self._gate)
            be.spmv_gated_dot(self.local, self.nl, x._ext_store, y, self._gate, out, fork)
            self.comm.exchange_forget()
            return True
        # below ~4 M local rows the plain SpMV + a separate dot is faster (measured per rank of
        # an 8-rank 256^3 run: 239 against 251 us per CG iteration; the fused kernel's partial
        # sums need two fold launches and its boundary-row share two more)
        if not (hasattr(be, "spmv_dot") and x.ld == 1 and y.ld == 1 and x.size[1] == 1 and
                self.local_format == "csr" and isinstance(self.local, Csr) and self.local.size[0] == self.local.size[1] and
                self.n_local >= self.fused_dot_min_rows):
            return False
        self.apply(x, y, dot_out=out)
        return True

    def can_start_with_reduce(self, x):
        """the halo exchange of apply(x, .) can be started together with an all-reduce
        (begin_exchange_with_reduce): device-resident communicator, zero-copy send planes"""
        comm = self.comm
        return (comm.size > 1 and self._side is not None and getattr(comm, "direct", False) and
                hasattr(comm, "all_reduce_exchange_begin") and hasattr(comm, "exchange_join") and
                self.send_displs is not None and x.ld == 1 and x.size[1] == 1)

    def begin_exchange_with_reduce(self, x, t):
        """start the all-reduce of t and the halo exchange of x on the side stream behind one
        fork; apply(x, y, started=True) must follow (its join ends both)"""
        gated = getattr(x, "_ext_halo", None) is not None and self._gate is not None
        recv = x._ext_halo if gated else self.recv_buf.values
        # (the product that follows opens the fork with its first wave where the communicator offers it)
        self._started_fork = self._fork_token() if gated and self.use_full_boundary else None
        self.comm.all_reduce_exchange_begin(t, recv, x.values, self.recv_counts,
                                            self.send_counts, self._side, self.send_displs)

    def apply(self, x, y, dot_out=None, started=False, join=True):
        """y_local = A[owned rows, :] x   (x, y: local parts, n_local x 1); started: the halo
        exchange of x is already under way (begin_exchange_with_reduce); join=False (with started,
        one-kernel product only): the main stream does NOT wait for the exchange's stream - whoever
        reads the all-reduced values next waits for this product's gate (HipBackend.step_gate);
        returns True in that case"""
        be, comm = self.backend, self.comm
        if started and self._gated(x, y):
            be.gate_open(self._side, self._gate)
            be.spmv_gated(self.local, self.nl, x._ext_store, y, self._gate,
                          getattr(self, "_started_fork", None))
            self._started_fork = None
            if not join:
                comm.exchange_forget()
                return True
            # (the all-reduce that was started with the exchange ends with the join)
            comm.exchange_join()
            return y
        if started:
            full = self.nl.get("full") if (self.use_full_boundary and hasattr(be, "rowlist_full")) else None
            if full is not None:
                be.spmv_rows(self.local, full["interior"][0], full["interior"][1], x, y)
                be.rowlist_full(self.nl, x, self.recv_buf, y, self._side)
                comm.exchange_join()
            else:
                be.spmv(self.local_op, x, y)
                comm.exchange_join()
                be.rowlist_add(self.nl, self.recv_buf, y)
            return y
        if dot_out is None and comm.size > 1 and self._gated(x, y):
            fork = self._fork_token()
            comm.exchange_begin(x._ext_halo, x.values, self.recv_counts, self.send_counts, self._side,
                                self.send_displs)
            be.gate_open(self._side, self._gate)
            be.spmv_gated(self.local, self.nl, x._ext_store, y, self._gate, fork)
            comm.exchange_forget()
            return y
        if dot_out is not None:
            local_spmv = lambda: be.spmv_dot(self.local, x, y, dot_out)
        else:
            local_spmv = lambda: be.spmv(self.local_op, x, y)
        direct = comm.size > 1 and self._side is not None and getattr(comm, "direct", False)
        zero_copy = direct and self.send_displs is not None and x.ld == 1 and x.size[1] == 1
        # 1. pack the rows the neighbours need (RowGatherer::apply_prepare)
        if not zero_copy:
            be.gather(x, self.send_idx, self.send_buf)
        # Contiguous partitions (one column): the boundary rows exist as COMPLETE rows over
        # [x | halo] (single-domain bits).  The local SpMV then covers the interior rows only and
        # the boundary rows are computed from the halo directly - with a device-resident
        # communicator on the exchange's stream, right behind the halo, overlapping the tail of
        # the local SpMV; otherwise after it in stream order.
        full = self.nl.get("full") if (dot_out is None and x.ld == 1 and y.ld == 1 and
                                       x.size[1] == 1 and hasattr(be, "rowlist_full") and
                                       self.use_full_boundary) else None
        if full is not None:
            local_spmv = lambda: be.spmv_rows(self.local, full["interior"][0], full["interior"][1], x, y)
        if full is not None and direct and hasattr(comm, "exchange_join"):
            if zero_copy:
                comm.exchange_begin(self.recv_buf.values, x.values, self.recv_counts,
                                    self.send_counts, self._side, self.send_displs)
            else:
                comm.exchange_begin(self.recv_buf.values, self.send_buf.values, self.recv_counts,
                                    self.send_counts, self._side)
            local_spmv()
            be.rowlist_full(self.nl, x, self.recv_buf, y, self._side)
            comm.exchange_join()
            return y
        if direct:
            # 2. exchange on a second stream (ordering by events inside the library),
            # overlapped with 3.
            if zero_copy:
                comm.exchange_begin(self.recv_buf.values, x.values, self.recv_counts,
                                    self.send_counts, self._side, self.send_displs)
            else:
                comm.exchange_begin(self.recv_buf.values, self.send_buf.values, self.recv_counts,
                                    self.send_counts, self._side)
            local_spmv()                                   # 3. local part
            comm.exchange_end()
        elif comm.size > 1 and self._side is not None and not comm.host_staging:
            # the same through torch.distributed
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                comm.all_to_all_v(self.recv_buf.values.view(-1), self.send_buf.values.view(-1),
                                  self.recv_counts, self.send_counts)
                done = torch.cuda.Event()
                done.record()
            local_spmv()                                   # 3. local part
            torch.cuda.current_stream().wait_event(done)
        else:
            if comm.size > 1 and not getattr(be, "is_host", False):
                be.synchronize()
            comm.all_to_all_v(self.recv_buf.values.view(-1), self.send_buf.values.view(-1),
                              self.recv_counts, self.send_counts)
            local_spmv()
        # 4. non-local part on the received halo (boundary rows only)
        if full is not None:
            be.rowlist_full(self.nl, x, self.recv_buf, y)
        elif dot_out is not None:
            be.rowlist_add_dot(self.nl, self.recv_buf, y, x, dot_out)
        else:
            be.rowlist_add(self.nl, self.recv_buf, y)
        return y


class DistributedCg:
    """Cg::apply_dense_impl (core/solver/cg.cpp:93-181) on distributed vectors:
    every dot / norm is the local kernel + an all-reduce of values that stay on
    the device (distributed/vector.cpp:473-592).

    Two latency measures for strong scaling, neither changes a result:
    * rho = <r,z> and ||r||^2 are reduced in ONE all-reduce of two values (the
      reference issues two);
    * the criterion check is asynchronous: its kernel marks stop_status on the
      device, cg::step_1/step_2 are masked by stop_status (cg_kernels.hpp), so
      the host enqueues `check_lag` further iterations before it reads the
      answer of iteration k (always exactly that many, on every rank) - they
      leave x, r, p untouched once the column has stopped.
      check_lag = 0 is the reference's lock-step behaviour."""

    def __init__(self, backend, comm, matrix, max_iters, reduction_factor=1e-10,
                 max_block_size=8, check_lag=None, fused=True, taped=True, fused_step_2=True):
        self.be, self.comm, self.a = backend, comm, matrix
        self.taped = bool(taped)
        self.fused_step_2 = bool(fused_step_2)
        self.max_iters, self.factor = int(max_iters), float(reduction_factor)
        self.m = backend.jacobi(matrix.local, max_block_size) if max_block_size else None
        self.num_iterations = 0
        self.residual_norm = None
        # never more checks in flight than the flag ring has slots
        self.check_lag = backend.max_check_lag if check_lag is None else \
            max(0, min(int(check_lag), 16 - 2))
        self.fused = bool(fused)
        import os
        self.step_1_check = os.environ.get("GKO_STEP1_CHECK", "1") != "0"
        n, dt = matrix.n_local, matrix.dtype
        self.r, self.z, self.p, self.q = (backend.vector(n, dt) for _ in range(4))
        if hasattr(matrix, "ext_vector"):
            self.p = matrix.ext_vector()     # the SpMV's input: halo room behind it (one-kernel product)
        self.beta, self.tau0 = backend.vector(1, dt), backend.vector(1, dt)
        # [rho, ||r||^2] pairs; the two pairs swap roles as rho / prev_rho
        self.pair_a = backend.scalar_pair(dt)
        self.pair_b = backend.scalar_pair(dt)
        self.flags, self.stop = backend.stop_flags()

    def _dot(self, x, y, out):
        self.be.local_dot(x, y, out)
        self.comm.all_reduce_sum_(out.values.view(-1))

    def _norm2(self, x, out):
        self.be.local_sqnorm(x, out)
        self.comm.all_reduce_sum_(out.values.view(-1))
        self.be.sqrt_(out)

    def _drain(self, pending, upto):
        """read (blocking) the checks of iterations <= upto, oldest first; the
        iteration that stopped, or None.  Which checks are read depends only on
        the iteration number, never on timing, so every rank leaves the loop in
        the same iteration and no collective is left unmatched."""
        while pending and pending[0][0] <= upto:
            it, token = pending.popleft()
            if self.be.check_done(token, True):
                return it
        return None

    def apply(self, b, x):
        from collections import deque
        be, a = self.be, self.a
        r, z, p, q = self.r, self.z, self.p, self.q
        beta = self.beta
        cur, prev = self.pair_a, self.pair_b
        be.cg_initialize(b, r, z, p, q, prev[1], cur[1], self.stop)
        a.apply(x, q)                          # r = b - A x
        neg = be.scalar(-1.0, b.dtype)
        r.add_scaled(neg, q)
        q.fill(0.0)
        self._norm2(b, self.tau0)              # ResidualNorm(rhs_norm) baseline
        pending = deque()
        fused = self.fused and hasattr(be, "cg_step_2_sqnorm")

        # have_sq: 0 nothing of the coming iteration exists yet, 1 ||r||^2 does (step_2 + norm),
        # 2 z, <r,z> and ||r||^2 do (step_2 + preconditioner in one kernel)
        def seg_a(cur, have_sq):
            pair, rho, tau = cur
            if have_sq < 2:
                if not (self.m is not None and fused and be.jacobi_apply_dot(self.m, r, z, rho)):
                    if self.m is not None:
                        self.m.apply(r, z)
                    else:
                        z.copy_from(r)
                    be.local_dot(r, z, rho)
                if not have_sq:
                    be.local_sqnorm(r, tau)
            self.comm.all_reduce_sum_(pair)    # one message: [<r,z>, ||r||^2]

        fused_s2 = fused and self.m is not None and hasattr(be, "cg_step_2_jacobi") and self.fused_step_2

        def seg_b(cur, prev, slot=None):
            if slot is not None:
                be.cg_step_1_check(p, z, cur[1], prev[1], cur[2], self.tau0, self.factor, self.stop, slot)
            else:
                be.cg_step_1(p, z, cur[1], prev[1], self.stop)
            if fused and hasattr(a, "apply_dot") and a.apply_dot(p, q, beta):
                self.comm.all_reduce_sum_(beta.values.view(-1))   # <p,q> came with the SpMV
            else:
                a.apply(p, q)
                self._dot(p, q, beta)
            # the pair that is `cur` in the next iteration receives <r_new, z_new> and ||r_new||^2
            if fused_s2 and be.cg_step_2_jacobi(self.m, x, r, p, q, beta, cur[1], self.stop, z,
                                                prev[1], prev[2]):
                return 2
            hs = fused and be.cg_step_2_sqnorm(x, r, p, q, beta, cur[1], self.stop, prev[2])
            if not hs:
                be.cg_step_2(x, r, p, q, beta, cur[1], self.stop)
            return 1 if hs else 0

        # The two halves of an iteration touch the same buffers every second
        # iteration (the [rho, tau] pairs alternate): each variant is recorded the
        # first time it runs and replayed afterwards (_lib.Tape).
        taped = self.taped and getattr(be, "tapeable", False) and getattr(self.comm, "tapeable", False)
        tapes = {}

        def run(key, fn, *args):
            if not taped:
                return fn(*args)
            t = tapes.get(key)
            if t is not None:
                return t.replay()
            with _record() as t:
                t.result = fn(*args)
            tapes[key] = t
            return t.result

        check_in_step_1 = (fused and self.step_1_check and hasattr(be, "cg_step_1_check") and
                           getattr(be, "check_takes_squared_norm", False) and
                           p.ld == 1 and z.ld == 1 and p.size[1] == 1)
        have_sq = 0
        it = -1
        while True:
            parity = (it + 1) & 1
            run(("a", parity, have_sq), seg_a, cur, have_sq)
            tau = cur[2]
            it += 1
            if it >= self.max_iters:
                stopped = self._drain(pending, it)
                if stopped is not None:
                    it = stopped
                break
            if check_in_step_1:
                # the criterion is the first thing cg::step_1's kernel does (gkoc_x_cg_step_1_check_*):
                # a column that has converged is left alone by it and by everything behind it, so
                # reading the answer AFTER the rest of the iteration is enqueued changes nothing
                tok = be.check_slot()
                pending.append((it, tok))
                have_sq = run(("b", parity, tok), seg_b, cur, prev, tok)
                stopped = self._drain(pending, it - self.check_lag)
                if stopped is not None:
                    it = stopped
                    break
                cur, prev = prev, cur
                continue
            if getattr(be, "check_takes_squared_norm", False):
                tok = be.check_begin(tau, self.tau0, self.factor, self.stop, squared=True)
            else:
                be.sqrt_(tau)
                tok = be.check_begin(tau, self.tau0, self.factor, self.stop)
            pending.append((it, tok))
            stopped = self._drain(pending, it - self.check_lag)
            if stopped is not None:
                it = stopped
                break
            have_sq = run(("b", parity), seg_b, cur, prev)
            cur, prev = prev, cur
        self.num_iterations = it
        if hasattr(self.a, "check_gate"):
            self.a.check_gate()
        return x


class DistributedPipeCg:
    """PipeCg::apply_dense_impl (core/solver/pipe_cg.cpp:95-297) on distributed vectors: the
    pipelined CG whose iteration needs ONE global reduction - the fit for a latency-bound
    strong-scaling run over xGMI.

    Per iteration (kernel sequence of the reference, same values):
      step_1:  x += t p, r -= t q, z -= t f, w -= t g          (t = rho / beta)
      dots:    rho = <r,z>, delta = <w,z>, ||r||^2   - ONE all-reduce of three values
      m = M^-1 w ; n = A m                            - runs WHILE the all-reduce travels
      check ; step_2: beta, p = z + s p, q = w + s q, f = m + s f, g = n + s g
    The reference computes the dots after n = A m; they only read r, z, w, which step_1
    finished, so here they are taken first (fused into step_1's pass on the device) and
    reduced on the side stream that also carries the halo exchange of n = A m, both in the
    same order on every rank.  rho / prev_rho alternate between two triples instead of being
    copied.  The criterion is ResidualNorm(rhs_norm) on ||r|| like DistributedCg's (its square
    rides in the same message), checked asynchronously with `check_lag` exactly as there:
    pipe_cg::step_1 / step_2 are masked by stop_status (pipe_cg_kernels.cpp:79-164)."""

    def __init__(self, backend, comm, matrix, max_iters, reduction_factor=1e-10,
                 max_block_size=8, check_lag=None, fused=True, taped=True, fused_steps=True,
                 fused_jacobi=True):
        self.be, self.comm, self.a = backend, comm, matrix
        self.taped = bool(taped)
        self.fused_steps = bool(fused_steps)
        self.fused_jacobi = bool(fused_jacobi)
        self.step_gate = os.environ.get("GKO_STEP_GATE", "1") != "0"

        self.max_iters, self.factor = int(max_iters), float(reduction_factor)
        self.m_op = backend.jacobi(matrix.local, max_block_size) if max_block_size else None
        self.num_iterations = 0
        self.check_lag = backend.max_check_lag if check_lag is None else \
            max(0, min(int(check_lag), 16 - 2))
        self.fused = bool(fused)
        n, dt = matrix.n_local, matrix.dtype
        (self.r, self.w, self.z, self.p, self.m, self.n, self.q, self.f, self.g) = (
            backend.vector(n, dt) for _ in range(9))
        if hasattr(matrix, "ext_vector"):
            self.m = matrix.ext_vector()     # the SpMV's input: halo room behind it (one-kernel product)
        self.beta, self.tau0 = backend.vector(1, dt), backend.vector(1, dt)
        self.beta2 = backend.vector(1, dt)     # the fused step_2 + step_1 reads one beta and writes the other
        # [rho, delta, ||r||^2]; the two triples swap roles as (rho, prev_rho)
        self.trip_a = backend.scalar_tuple(3, dt)
        self.trip_b = backend.scalar_tuple(3, dt)
        self.flags, self.stop = backend.stop_flags()
        self._side = matrix._side

    def _precond(self, src, dst):
        if self.m_op is not None:
            self.m_op.apply(src, dst)
        else:
            dst.copy_from(src)

    def _drain(self, pending, upto):
        while pending and pending[0][0] <= upto:
            it, token = pending.popleft()
            if self.be.check_done(token, True):
                return it
        return None

    def _check_begin(self, tau):
        be = self.be
        if getattr(be, "check_takes_squared_norm", False):
            return be.check_begin(tau, self.tau0, self.factor, self.stop, squared=True)
        be.sqrt_(tau)
        return be.check_begin(tau, self.tau0, self.factor, self.stop)

    def apply(self, b, x):
        from collections import deque
        be, a, comm = self.be, self.a, self.comm
        r, w, z, p, m, n, q, f, g = (self.r, self.w, self.z, self.p, self.m, self.n, self.q,
                                     self.f, self.g)
        beta = self.beta
        cur, prev = self.trip_a, self.trip_b
        fused = self.fused and hasattr(be, "pipe_cg_step_1_dots")
        be.pipe_cg_initialize_1(b, r, prev[1][0], self.stop)   # r = b ; prev_rho = 1
        a.apply(x, q)                                           # r = b - A x
        r.add_scaled(be.scalar(-1.0, b.dtype), q)
        self._precond(r, z)
        a.apply(z, w)
        self._precond(w, m)
        a.apply(m, n)
        # ResidualNorm(rhs_norm) baseline
        be.local_sqnorm(b, self.tau0)
        comm.all_reduce_sum_(self.tau0.values.view(-1))
        be.sqrt_(self.tau0)
        be.local_dot(r, z, cur[1][0])
        be.local_dot(w, z, cur[1][1])
        be.local_sqnorm(r, cur[1][2])
        comm.all_reduce_sum_(cur[0])
        pending = deque()
        it = 0
        pending.append((it, self._check_begin(cur[1][2])))
        if self._drain(pending, it) is not None:
            self.num_iterations = 0
            return x
        be.pipe_cg_initialize_2(p, q, f, g, beta, z, w, m, n, cur[1][1])

        def seg(cur, prev):
            """one iteration up to the all-reduced scalars; `prev` receives the new triple"""
            out, (rho_new, delta_new, tau_new) = prev
            rho = cur[1][0]
            if not (fused and be.pipe_cg_step_1_dots(x, r, z, w, p, q, f, g, rho, beta, self.stop,
                                                     out)):
                be.pipe_cg_step_1(x, r, z, w, p, q, f, g, rho, beta, self.stop)
                be.local_dot(r, z, rho_new)
                be.local_dot(w, z, delta_new)
                be.local_sqnorm(r, tau_new)
            comm.all_reduce_begin(out, self._side)      # one message: [rho, delta, ||r||^2]
            self._precond(w, m)
            a.apply(m, n)
            comm.all_reduce_end()

        def seg_step2(cur, prev):
            # cur = the new triple, prev = the old one (prev_rho)
            be.pipe_cg_step_2(beta, p, q, f, g, z, w, m, n, prev[1][0], cur[1][0], cur[1][1],
                              self.stop)

        taped = self.taped and getattr(be, "tapeable", False) and getattr(comm, "tapeable", False)
        tapes = {}

        def run(key, fn, *args):
            if not taped:
                return fn(*args)
            t = tapes.get(key)
            if t is not None:
                return t.replay()
            with _record() as t:
                t.result = fn(*args)
            tapes[key] = t
            return t.result

        # step_2 of iteration k and step_1 + the three dots of iteration k + 1 as ONE kernel
        # (gkoc_x_pipe_cg_step_2_step_1_dots_*: 144 instead of 192 bytes per row, one launch less);
        # the all-reduce of its three values still travels while m = M^-1 w and n = A m run.  Same
        # vectors bit for bit; beta alternates between two scalars (the kernel reads one, writes
        # the other).
        if fused and self.fused_steps and hasattr(be, "pipe_cg_step_2_step_1_dots") and \
                all(v.ld == 1 and v.size[1] == 1 for v in (x, r, z, w, p, q, f, g, m, n)):
            betas = (beta, self.beta2)
            # with block-Jacobi in its fast-path layout the step kernel also applies the
            # preconditioner (m = M w from the registers that hold the new w)
            with_m = self.m_op is not None and self.fused_jacobi and \
                hasattr(be, "pipe_cg_steps_jacobi") and \
                be.pipe_cg_steps_jacobi(self.m_op, x, r, z, w, p, q, f, g, m, n, None, None, None, None,
                                        None, None, None, probe=True)

            def head(cur, prev):                 # the very first step_1 (+ dots): nothing to fuse it with
                out = prev[0]
                if not be.pipe_cg_step_1_dots(x, r, z, w, p, q, f, g, cur[1][0], betas[0], self.stop, out):
                    raise GkoError("DistributedPipeCg: fused step kernels unavailable for this layout")
                if with_m:
                    self._precond(w, m)

            # m is final where the three sums are (the step kernel computed it): reduction and halo
            # exchange start together behind one fork, the SpMV's join ends both
            together = with_m and hasattr(a, "can_start_with_reduce") and a.can_start_with_reduce(m)

            # ... and where the product is the one-kernel one, nothing joins: the step kernel waits for
            # the product's gate (set behind the reduction) and judges the criterion itself
            no_join = (together and self.step_gate and hasattr(be, "step_gate") and
                       hasattr(a, "_gated") and a._gated(m, n) and
                       getattr(be, "check_takes_squared_norm", False))

            def mid(prev):                       # reduce what the last step kernel left in `prev`
                if together:
                    a.begin_exchange_with_reduce(m, prev[0])
                    a.apply(m, n, started=True, join=not no_join)
                    return
                comm.all_reduce_begin(prev[0], self._side)
                if not with_m:
                    self._precond(w, m)
                a.apply(m, n)
                comm.all_reduce_end()

            def tail(cur, prev, b_in, b_out, slot=None):    # step_2 (prev_rho = prev, rho / delta = cur) + next step_1
                if with_m:
                    sg = be.step_gate(a._gate, cur[1][2], self.tau0, self.factor, self.stop, slot) \
                        if slot is not None else None
                    be.pipe_cg_steps_jacobi(self.m_op, x, r, z, w, p, q, f, g, m, n, prev[1][0], cur[1][0],
                                            cur[1][1], b_in, b_out, self.stop, prev[0], gate=sg)
                else:
                    be.pipe_cg_step_2_step_1_dots(x, r, z, w, p, q, f, g, m, n, prev[1][0], cur[1][0],
                                                  cur[1][1], b_in, b_out, self.stop, prev[0])

            run(("h",), head, cur, prev)
            while True:
                parity = it & 1
                run(("m", parity), mid, prev)
                cur, prev = prev, cur
                it += 1
                if it >= self.max_iters:
                    stopped = self._drain(pending, it)
                    if stopped is not None:
                        it = stopped
                    break
                if no_join:
                    # the criterion is the first thing the step kernel does (behind its wait for the
                    # reduced values): a column that has converged is left alone by it, so looking at
                    # the answer after the kernel is enqueued changes nothing
                    tok = be.check_slot()
                    pending.append((it, tok))
                    run(("t", parity, tok), tail, cur, prev, betas[parity], betas[1 - parity], tok)
                    stopped = self._drain(pending, it - self.check_lag)
                    if stopped is not None:
                        it = stopped
                        break
                    continue
                pending.append((it, self._check_begin(cur[1][2])))
                stopped = self._drain(pending, it - self.check_lag)
                if stopped is not None:
                    it = stopped
                    break
                run(("t", parity), tail, cur, prev, betas[parity], betas[1 - parity])
            self.num_iterations = it
            if hasattr(self.a, "check_gate"):
                self.a.check_gate()
            return x

        while True:
            parity = it & 1
            run(("s", parity), seg, cur, prev)
            cur, prev = prev, cur
            it += 1
            if it >= self.max_iters:
                stopped = self._drain(pending, it)
                if stopped is not None:
                    it = stopped
                break
            pending.append((it, self._check_begin(cur[1][2])))
            stopped = self._drain(pending, it - self.check_lag)
            if stopped is not None:
                it = stopped
                break
            run(("t", parity), seg_step2, cur, prev)
        self.num_iterations = it
        if hasattr(self.a, "check_gate"):
            self.a.check_gate()
        return x


class _LocalOperator:
    """what the Krylov drivers of solver.py need from a system matrix: the local
    part of a DistributedMatrix behaves like a square LinOp on local vectors"""

    def __init__(self, dm):
        self.dm = dm
        self.exec = dm.backend.exec
        self.size = (dm.n_local, dm.n_local)

    def get_size(self):
        return self.size

    def apply(self, *args):
        if len(args) != 2:
            raise GkoError("distributed operator: only y = A x")
        return self.dm.apply(*args)


def DistributedGmres(backend, comm, matrix, max_iters, reduction_factor=1e-10,
                     max_block_size=8, krylov_dim=100, ortho_method="mgs"):
    """Restarted GMRES on a row-partitioned matrix: solver.Gmres (the driver of
    core/solver/gmres.cpp:321-621) with every dot / norm all-reduced
    (distributed/vector.cpp:473-592) and the block-Jacobi preconditioner built
    from the local diagonal block.  The small Hessenberg / Givens kernels run
    replicated on every rank on identical, already reduced inputs.
    HipBackend only."""
    from . import solver as _solver
    from . import stop as _stop
    f = (_solver.Gmres.build().with_krylov_dim(krylov_dim).with_ortho_method(ortho_method)
         .with_criteria(_stop.Iteration.build().with_max_iters(max_iters),
                        _stop.ResidualNorm.build().with_reduction_factor(reduction_factor)))
    if max_block_size:
        f = f.with_generated_preconditioner(backend.jacobi(matrix.local, max_block_size))
    s = f.on(backend.exec).generate(_LocalOperator(matrix))
    s._comm = comm
    return s


# --------------------------------------------------------------- bench helper
class SlabPartition(Partition):
    def __init__(self, grid, n_parts):
        p = Partition.build_slabs(grid, n_parts)
        super().__init__(p.offsets)
        self.grid = grid
        planes = Partition.build_from_global_size_uniform(n_parts, grid)
        self.plane_offsets = planes.offsets


class DistributedStencil:
    """27-pt grid^3 Laplacian, z-slab partitioned, generated on each GPU
    (benchmark/utils/stencil_matrix.hpp semantics, see matrix.stencil_csr)."""

    def __init__(self, exec_, part, rank, comm=None):
        self.exec, self.part, self.rank = exec_, part, rank
        self.comm = comm or default_comm(exec_, halo_elems=part.grid ** 2)
        z0, z1 = part.plane_offsets[rank], part.plane_offsets[rank + 1]
        owned = stencil_csr(exec_, 3, part.grid, z0=z0, nz=z1 - z0)
        self.backend = HipBackend(exec_)
        self.matrix = DistributedMatrix(self.backend, self.comm, part, owned)
        self.n_local = self.matrix.n_local
        nnz = torch.tensor([owned.get_num_stored_elements()], dtype=torch.int64,
                           device=exec_.device)
        self.comm.all_reduce_sum_(nnz)
        self.global_nnz = int(nnz.item())

    def random_vector(self, seed):
        lo, hi = self.part.range_of(self.rank)
        full = np.random.default_rng(seed).uniform(-1, 1, self.part.n_global)
        if hasattr(self.matrix, "ext_vector"):
            # halo room behind the vector: its products run as one kernel (DistributedMatrix.apply)
            v = self.matrix.ext_vector()
            v.values.copy_(torch.from_numpy(np.ascontiguousarray(full[lo:hi])).view(-1, 1))
            return v
        return Dense.from_numpy(self.exec, full[lo:hi])

    def zeros_vector(self):
        return Dense.create(self.exec, (self.n_local, 1)).fill(0.0)

    def apply(self, x, y):
        return self.matrix.apply(x, y)

    def profile(self, x, y, steps=20):
        """device time of the pieces of one distributed apply on THIS rank (events on the
        executor's stream): the local block alone and the boundary rows alone; the bench
        line carries them next to the whole apply so that an N-GPU record shows where the
        time of a rank goes (the exchange latency is in comm_self_check's numbers)"""
        m, be = self.matrix, self.backend
        out = {}
        for name, fn in (("local_spmv_ms", lambda: be.spmv(m.local, x, y)),
                         ("boundary_rows_ms", lambda: be.rowlist_add(m.nl, m.recv_buf, y))):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                fn()
            e1.record()
            e1.synchronize()
            out[name] = round(e0.elapsed_time(e1) / steps, 4)
        out["n_local_rows"], out["n_halo"], out["n_boundary_rows"] = m.n_local, m.n_halo, m.nl["n"]
        return out

    def prepare_cg(self, iters, barrier):
        """set-up + one warm-up solve; returns the set-up time"""
        import time
        t0 = time.perf_counter()
        self._cg = DistributedCg(self.backend, self.comm, self.matrix, iters, 1e-30, 8)
        barrier()
        t_setup = time.perf_counter() - t0
        self._rhs = Dense.create(self.exec, (self.n_local, 1)).fill(1.0)
        self._sol = self.zeros_vector()
        self._cg.apply(self._rhs, self._sol)
        barrier()
        return t_setup

    def prepare_pipe_cg(self, iters, barrier):
        """the same for the pipelined CG (one all-reduce per iteration)"""
        self._pcg = DistributedPipeCg(self.backend, self.comm, self.matrix, iters, 1e-30, 8)
        self._pcg.apply(self._rhs, self._sol.fill(0.0))
        barrier()

    def timed_pipe_cg(self, barrier):
        import time
        self._sol.fill(0.0)
        barrier()
        t1 = time.perf_counter()
        self._pcg.apply(self._rhs, self._sol)
        barrier()
        return self._pcg.num_iterations, time.perf_counter() - t1

    def timed_cg(self, barrier):
        import time
        self._sol.fill(0.0)
        barrier()
        t1 = time.perf_counter()
        self._cg.apply(self._rhs, self._sol)
        barrier()
        return self._cg.num_iterations, time.perf_counter() - t1
