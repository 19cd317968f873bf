"""Worker of tests/test_comm_mailbox_gpu.py (launched with torch.distributed.run, all ranks on cuda:0):
the collectives of the mailbox transport (csrc/comm_ipc.hpp, gkoc_comm_ipc_*) against their definition -
all-reduce: the sum in RANK ORDER (numpy, sequential), the same bits on every rank; exchange: every
segment where MPI_Alltoallv would put it (include/ginkgo/core/base/mpi.hpp:838, :1441)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def gather_host(arr):
    """every rank's numpy array, in rank order (gloo)"""
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, arr)
    return out


def rank_order_sum(parts):
    s = parts[0].copy()
    for p in parts[1:]:
        s = s + p            # one rounding per rank, in rank order: what all_reduce_kernel does
    return s


def main():
    mode = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import ginkgo_amd as g
    import ginkgo_amd.distributed as gd
    from ginkgo_amd._lib import call, lib, GkoError

    ex = g.Cdna4Executor.create(0)
    dev = ex.device
    comm = gd.IpcComm(ex, slot_bytes=1 << 20)
    side = torch.cuda.Stream(device=dev)
    rng = np.random.default_rng(1000 + rank)
    report = {"ranks": world, "window_uncached": comm.window_uncached}

    # ---- all-reduce: 1 .. 100 values (more than one launch beyond 32), doubles and floats ----------
    for dtype, npdt in ((torch.float64, np.float64), (torch.float32, np.float32)):
        for n in (1, 2, 3, 31, 32, 33, 100):
            mine = (rng.uniform(-1, 1, n) * 10.0 ** rng.integers(-8, 8, n)).astype(npdt)
            t = torch.from_numpy(mine).to(dev)
            comm.all_reduce_sum_(t)
            want = rank_order_sum(gather_host(mine))
            got = t.cpu().numpy()
            assert got.tobytes() == want.tobytes(), (rank, n, dtype, got, want)
    # the overlapped form on the side stream, behind a kernel that is late
    for rep in range(20):
        mine = rng.uniform(-1, 1, 3)
        t = torch.from_numpy(mine).to(dev)
        if rep % 3 == rank % 3:
            call("gkoc_debug_delay", ex.stream, C.c_int64(300), 1, 64, 0)      # this rank's values come late
        comm.all_reduce_begin(t, side)
        comm.all_reduce_end()
        want = rank_order_sum(gather_host(mine))
        ex.synchronize()
        assert t.cpu().numpy().tobytes() == want.tobytes(), (rank, rep, hex(comm.status()), t.cpu().tolist(), want.tolist())
    # ---- many in a row with values that change every time (parities, epochs): checked at the end ----
    iters = 2000
    acc = torch.zeros(2, dtype=torch.float64, device=dev)
    t = torch.empty(2, dtype=torch.float64, device=dev)
    base = torch.tensor([rank + 1.0, 0.25 * (rank + 1)], dtype=torch.float64, device=dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(iters):
        torch.mul(base, float(k % 7 + 1), out=t)
        comm.all_reduce_sum_(t)
        acc += t
    torch.cuda.synchronize(dev)
    report["all_reduce_chain_us"] = round((time.perf_counter() - t0) / iters * 1e6, 2)
    tot = world * (world + 1) / 2
    ksum = sum(k % 7 + 1 for k in range(iters))
    assert acc.cpu().tolist() == [tot * ksum, 0.25 * tot * ksum], acc.cpu().tolist()

    # ---- exchange: the halo pattern of a slab partition, zero-copy out of the vector (displacements) ----
    plane = 4099                               # odd: segments that are only 8-byte aligned
    nloc = 6 * plane
    x = torch.from_numpy(rng.uniform(-1, 1, nloc)).to(dev)
    xs = gather_host(x.cpu().numpy())
    peers = [p for p in (rank - 1, rank + 1) if 0 <= p < world]
    counts = [plane if p in peers else 0 for p in range(world)]
    displs = [0 if p == rank - 1 else (nloc - plane if p == rank + 1 else 0) for p in range(world)]
    recv = torch.zeros(plane * len(peers), dtype=torch.float64, device=dev)
    for rep in range(5):
        recv.zero_()
        comm.exchange_begin(recv, x, counts, counts, side, displs)
        comm.exchange_end()
        got = recv.cpu().numpy()
        for i, p in enumerate(peers):
            want = xs[p][nloc - plane:] if p == rank - 1 else xs[p][:plane]
            assert np.array_equal(got[i * plane:(i + 1) * plane], want), (rank, p, rep)
    # ---- all-to-all with random counts (zeros, self messages, 4-byte values, odd lengths) ----
    for rep in range(6):
        cm = np.random.default_rng(77 + rep).integers(0, 5000, (world, world))      # cm[s, d]: s -> d
        cm[np.random.default_rng(5 + rep).uniform(size=cm.shape) < 0.3] = 0
        sc, rc = cm[rank].tolist(), cm[:, rank].tolist()
        send = torch.from_numpy(rng.uniform(-1, 1, max(sum(sc), 1)).astype(np.float32)).to(dev)
        recv = torch.full((max(sum(rc), 1),), -7.0, dtype=torch.float32, device=dev)
        comm.exchange_begin(recv, send, rc, sc, side if rep % 2 else None)
        comm.exchange_end()
        sends = gather_host(send.cpu().numpy())
        got, pos = recv.cpu().numpy(), 0
        for s in range(world):
            off = int(cm[s, :rank].sum())
            assert np.array_equal(got[pos:pos + cm[s, rank]], sends[s][off:off + cm[s, rank]]), (rank, s, rep)
            pos += int(cm[s, rank])
    # ---- the byte form (MPI_Alltoallv of the MPI layer): offsets on both sides, one-byte granularity ----
    sb = [(7 * (rank + 1) + 13 * p) % 97 for p in range(world)]
    rb = [(7 * (p + 1) + 13 * rank) % 97 for p in range(world)]
    so = [int(np.sum(sb[:p])) + 3 * p for p in range(world)]          # gaps between the segments
    ro = [int(np.sum(rb[:p])) + 5 * p for p in range(world)]
    send8 = torch.from_numpy(rng.integers(0, 255, so[-1] + sb[-1] + 8).astype(np.uint8)).to(dev)
    recv8 = torch.zeros(ro[-1] + rb[-1] + 8, dtype=torch.uint8, device=dev)
    arr = lambda v: (C.c_int64 * world)(*v)
    call("gkoc_comm_all_to_all_v_bytes", comm._handle, ex.stream, send8, arr(sb), arr(so), recv8, arr(rb), arr(ro))
    ex.synchronize()
    sends, sos = gather_host(send8.cpu().numpy()), gather_host(so)
    got = recv8.cpu().numpy()
    for s in range(world):
        assert np.array_equal(got[ro[s]:ro[s] + rb[s]], sends[s][sos[s][rank]:sos[s][rank] + rb[s]]), (rank, s)
    # ---- one-way traffic with a slow receiver: the sender may be two messages ahead, never three ----
    if world >= 2:
        n1 = 3000
        seq_ok = True
        sc = [n1 if (rank == 0 and p == 1) else 0 for p in range(world)]
        rc = [n1 if (rank == 1 and p == 0) else 0 for p in range(world)]
        buf = torch.zeros(n1, dtype=torch.float64, device=dev)
        seen = torch.zeros(40, dtype=torch.float64, device=dev)
        for k in range(40):
            if rank == 0:
                buf.fill_(float(k + 1))
            if rank == 1 and k % 4 == 0:
                call("gkoc_debug_delay", ex.stream, C.c_int64(2000), 1, 64, 0)     # 2 ms behind
            comm.exchange_begin(buf, buf, rc, sc, None)
            comm.exchange_end()
            if rank == 1:
                seen[k] = buf[0] + buf[-1]
        ex.synchronize()
        if rank == 1:
            seq_ok = seen.cpu().tolist() == [2.0 * (k + 1) for k in range(40)]
        assert seq_ok, seen.cpu().tolist()
    # ---- a message larger than the slot is refused, not truncated ----
    big = (1 << 20) // 8 + 1
    huge = torch.zeros(big, dtype=torch.float64, device=dev)
    cnt = [big if p == rank else 0 for p in range(world)]
    try:
        comm.exchange_begin(huge, huge, cnt, cnt, None)
        raise AssertionError("a message larger than the slot was accepted")
    except GkoError as e:
        assert "larger than the window" in str(e), str(e)
    # ---- latencies (what bench.py reports) and the status word ----
    chk = gd.comm_self_check(ex, comm, n_elems=65536, reps=50)
    report.update({k: chk[k] for k in ("all_reduce_us", "all_reduce_overlapped_us", "exchange_us")})
    comm.check()
    assert comm.status() == 0
    # ---- patience (gkoc_comm_set_patience_ms): a peer that is late beyond it is REPORTED, the kernel
    # ends instead of waiting; the late rank itself finds every word there.  The last operation of this
    # communicator: what the impatient rank computed is not a sum.
    if world == 2:
        comm.set_patience_ms(200)
        t = torch.ones(2, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        dist.barrier()
        if rank == 0:
            time.sleep(1.5)
        comm.all_reduce_sum_(t)
        torch.cuda.synchronize(dev)
        st = comm.status()
        if rank == 0:
            assert st == 0 and t.cpu().tolist() == [2.0, 2.0], (hex(st), t.cpu().tolist())
        else:
            assert st & 1, hex(st)
        report["impatient_status"] = st
        comm.set_patience_ms(0)
    comm.close()
    dist.barrier()
    if rank == 0:
        print("IPC_REPORT " + json.dumps(report))
        print(f"ipc_worker OK mode={mode} world={world}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
