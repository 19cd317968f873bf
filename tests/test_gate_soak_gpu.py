"""Soak of the two places where a kernel of the Krylov loop WAITS for another stream's work instead of
being ordered behind it by an event:
  * the one-kernel distributed product (csrc/csr_spmv_pipe.hpp, GATE): its boundary waves poll the gate
    word with a relaxed load and pay the agent-scope acquire only if they had to wait;
  * the fused PipeCg step kernel (csrc/fused.hpp step_gate_enter): waits for the gate, then reads the
    all-reduced scalars.
Thousands of consecutive iterations on one GPU with the protocol of the real path - the exchange's
stream is forked behind the main stream by the product's own first wave / a fork kernel, a kernel
spread over all XCDs REWRITES the halo (the scalars) with values that depend on the iteration, the gate
opens early or late at random - and every iteration's result is compared on the device with the same
product through the stream-ordered kernels.  One stale cache line anywhere shows as a mismatch.
Reference behaviour to match: the halo is complete before the non-local part is applied,
core/distributed/matrix.cpp:476-492 (req.wait() in front of the non-local apply)."""
import ctypes as C
import os

import numpy as np
import pytest

from util import record_perf

pytestmark = pytest.mark.gpu

ITERS = int(os.environ.get("GKO_SOAK_ITERS", "10000"))


def _slab(gexec, grid, world, rank):
    import ginkgo_amd as g
    import ginkgo_amd.distributed as gd
    part = gd.SlabPartition(grid, world)
    lo, hi = part.range_of(rank)
    z0, z1 = part.plane_offsets[rank], part.plane_offsets[rank + 1]
    owned = g.stencil_csr(gexec, 3, grid, z0=z0, nz=z1 - z0)
    be = gd.HipBackend(gexec)
    local, nl, recv_gidx = be.split(owned, lo, hi, grid ** 3)
    return be, local, nl, recv_gidx, lo, hi


@pytest.mark.parametrize("grid,world,rank,iters", [(128, 8, 3, ITERS), (256, 8, 3, ITERS // 5), (64, 64, 7, ITERS // 2)])
def test_gated_product_soak(gexec, grid, world, rank, iters):
    """(128, 8): 16-plane slabs, 512 boundary waves; (256, 8): the per-rank slab of the 8-GPU headline
    run, 2048 boundary waves; (64, 64): a one-plane slab - every wave is a boundary wave"""
    import time
    import torch
    import ginkgo_amd as g
    from ginkgo_amd._lib import call
    be, local, nl, recv_gidx, lo, hi = _slab(gexec, grid, world, rank)
    f = nl["full"]
    assert f["gated"]
    n_loc, n_halo = hi - lo, recv_gidx.numel()
    dev = gexec.device
    rng = np.random.default_rng(grid + rank)
    store = gexec.zeros((f["halo_base"] + n_halo,), torch.float64)
    store[:n_loc] = torch.from_numpy(rng.uniform(-1, 1, n_loc)).to(dev)
    halo_live = store[f["halo_base"]:f["halo_base"] + n_halo]
    base = torch.from_numpy(rng.uniform(-1, 1, n_halo)).to(dev)
    xl = g.Dense(gexec, store[:n_loc].clone().view(-1, 1))          # the same local vector for the reference
    hv = g.Dense(gexec, torch.zeros(n_halo, 1, dtype=torch.float64, device=dev))
    y, y2 = be.vector(n_loc), be.vector(n_loc)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    gate = be.gate_new()
    side = be.side_stream()
    sst = C.c_void_p(side.cuda_stream)
    mst = gexec.stream
    word = gexec.zeros((64,), torch.int32)
    choice = np.random.default_rng(5).integers(0, 6, iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1, iters + 1):
        scale = 1.0 + (k % 251) * 2.0 ** -9                          # exact in double: halo_k = base * scale
        how = int(choice[k - 1])
        # side: behind the main stream's work so far (the product's own first wave opens the fork) ...
        if how == 2:
            call("gkoc_stream_fork", mst, sst, word, C.c_uint32(k))  # ... (a fork kernel of its own, see below)
        else:
            call("gkoc_stream_fork_wait", sst, word, C.c_uint32(k))
        if how == 0:
            call("gkoc_debug_delay", sst, C.c_int64(200), 1, 64, 0)          # the exchange is late
        elif how == 1:
            call("gkoc_debug_delay", sst, C.c_int64(30), 64, 512, 32768)     # ... through RCCL-sized kernels
        with torch.cuda.stream(side):
            torch.mul(base, scale, out=halo_live)                    # the "exchange": a kernel on every XCD
        be.gate_open(side, gate)
        if how == 2:
            call("gkoc_debug_delay", mst, C.c_int64(150), 1, 64, 0)          # the product is late: the gate has been open for long
        be.spmv_gated(local, nl, store, y, gate, fork=(word, C.c_uint32(k)))
        # the reference: stream-ordered kernels on the main stream, halo in a buffer of its own
        torch.mul(base, scale, out=hv.values.view(-1))
        be.spmv_rows(local, f["interior"][0], f["interior"][1], xl, y2)
        be.rowlist_full(nl, xl, hv, y2)
        bad += (y.values != y2.values).sum()
        if k % 1000 == 0:
            assert int(bad.item()) == 0, f"iteration <= {k}: {int(bad.item())} entries differ"
            assert int(gate[0][1].item()) == 0, "a boundary wave gave up waiting"
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert int(bad.item()) == 0 and int(gate[0][1].item()) == 0
    assert int(word[0].item()) == iters
    # negative control - the comparison does see a stale halo: the gate opened BEFORE the halo is rewritten
    k = iters + 1
    call("gkoc_stream_fork", mst, sst, word, C.c_uint32(k))
    be.gate_open(side, gate)
    call("gkoc_debug_delay", sst, C.c_int64(2000), 1, 64, 0)
    with torch.cuda.stream(side):
        torch.mul(base, 3.0, out=halo_live)
    be.spmv_gated(local, nl, store, y, gate)
    torch.mul(base, 3.0, out=hv.values.view(-1))
    be.spmv_rows(local, f["interior"][0], f["interior"][1], xl, y2)
    be.rowlist_full(nl, xl, hv, y2)
    torch.cuda.synchronize()
    assert int((y.values != y2.values).sum().item()) > 0, "the soak's comparison cannot see a stale halo"
    record_perf("gated_product_soak", grid=grid, world=world, rank=rank, iters=iters, seconds=round(el, 2),
                boundary_waves=(f["head"] + f["tail"] + 63) // 64)


def test_step_gate_soak(gexec):
    """the fused PipeCg step kernel behind its gate: the three all-reduced scalars are rewritten on the side
    stream every iteration (iteration-dependent), the gate opens early or late; ten vectors restored,
    stepped with the gate and - from a copy - with the stream-ordered criterion + plain step kernel; the
    partial sums and x must agree bit for bit every time"""
    import time
    import torch
    import ginkgo_amd as g
    import ginkgo_amd.distributed as gd
    from ginkgo_amd._lib import call
    iters = max(ITERS // 4, 10)
    grid = 24
    n = grid ** 3
    be = gd.HipBackend(gexec)
    dev = gexec.device
    a = g.stencil_csr(gexec, 3, grid)
    m_op = be.jacobi(a, 8)
    rng = np.random.default_rng(3)
    base = torch.from_numpy(rng.uniform(-1, 1, (10, n))).to(dev)

    def vectors():
        store = torch.empty(10, n, dtype=torch.float64, device=dev)
        return store, [g.Dense(gexec, store[i].view(-1, 1)) for i in range(10)]

    sa, va = vectors()
    sb, vb = vectors()
    trip_a_t, trip_a = be.scalar_tuple(3)
    trip_b_t, trip_b = be.scalar_tuple(3)
    prev_rho, tau0 = be.scalar(0.7), be.scalar(4.0)
    b_in = be.scalar(0.4)
    bo_a, bo_b = be.scalar(0.0), be.scalar(0.0)
    out_a, out_b = gexec.zeros((3,), torch.float64), gexec.zeros((3,), torch.float64)
    _, stop_a = be.stop_flags()
    _, stop_b = be.stop_flags()
    slot_a, slot_b = be.check_slot(), be.check_slot()
    gate = be.gate_new()
    side = be.side_stream()
    sst, mst = C.c_void_p(side.cuda_stream), gexec.stream
    word = gexec.zeros((64,), torch.int32)
    sg = be.step_gate(gate, trip_a[2], tau0, 1e-10, stop_a, slot_a)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    src = torch.tensor([1.3, 0.9, 2.0], dtype=torch.float64, device=dev)
    choice = np.random.default_rng(9).integers(0, 4, iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1, iters + 1):
        scale = 1.0 + (k % 127) * 2.0 ** -8
        sa.copy_(base)
        sb.copy_(base)
        stop_a.zero_()
        stop_b.zero_()
        call("gkoc_stream_fork", mst, sst, word, C.c_uint32(k))      # side: behind the restores
        if choice[k - 1] == 0:
            call("gkoc_debug_delay", sst, C.c_int64(100), 1, 64, 0)
        with torch.cuda.stream(side):
            torch.mul(src, scale, out=trip_a_t)                      # the "all-reduce" result arrives
        be.gate_open(side, gate)
        if choice[k - 1] == 1:
            call("gkoc_debug_delay", mst, C.c_int64(100), 1, 64, 0)
        x, r, z, w, p, q, ff, gg, m, nn = va
        assert be.pipe_cg_steps_jacobi(m_op, x, r, z, w, p, q, ff, gg, m, nn, prev_rho, trip_a[0], trip_a[1],
                                       b_in, bo_a, stop_a, out_a, gate=sg)
        torch.mul(src, scale, out=trip_b_t)
        call("gkoc_implicit_residual_norm_f64", mst, 1, trip_b[2].values, tau0.values, C.c_double(1e-10),
             C.c_uint8(2), C.c_int(1), stop_b, be._chk_host[slot_b], None, None)
        x, r, z, w, p, q, ff, gg, m, nn = vb
        assert be.pipe_cg_steps_jacobi(m_op, x, r, z, w, p, q, ff, gg, m, nn, prev_rho, trip_b[0], trip_b[1],
                                       b_in, bo_b, stop_b, out_b)
        bad += (out_a != out_b).sum() + (sa != sb).sum() + (bo_a.values != bo_b.values).sum()
        if k % 500 == 0:
            assert int(bad.item()) == 0, f"iteration <= {k}: {int(bad.item())} values differ"
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert int(bad.item()) == 0 and int(gate[0][1].item()) == 0
    record_perf("step_gate_soak", iters=iters, seconds=round(el, 2))
