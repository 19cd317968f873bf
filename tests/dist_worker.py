"""Worker for the multi-rank tests (launched with torch.distributed.run).
mode cpu : gloo + OracleBackend (no GPU)   - checks the distributed logic
mode gpu : gloo + HipBackend, all ranks on cuda:0 (host-staged exchange)
Each rank compares its slice of the distributed SpMV / CG result with the
single-process oracle result of the full problem; exits non-zero on mismatch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def fallback_case(mode):
    """default_comm(): the bring-up of the library's RCCL communicator fails on ONE rank (injected
    through GKO_COMM_INJECT_FAIL, stage named in the mode) - every rank must end up on
    torch.distributed together, with no rank left inside a collective, and the result of a
    distributed apply must be the usual one."""
    import types
    import ginkgo_amd.distributed as gd
    rank, world = dist.get_rank(), dist.get_world_size()
    if mode.endswith("cpu"):
        ex = types.SimpleNamespace(device=torch.device("cpu"), stream=None)
    else:
        import ginkgo_amd as g
        ex = g.Cdna4Executor.create(0)
    # both device-resident transports, each attempted although the group is gloo: the library's mailboxes
    # (IpcComm; on a box without a GPU its window cannot be created on ANY rank) and RCCL
    os.environ["GKO_COMM"] = "ipc"
    comm = gd.default_comm(ex)
    assert type(comm) is gd.TorchComm and gd.default_comm.last["chosen"] == "TorchComm", (type(comm), gd.default_comm.last)
    os.environ["GKO_COMM"] = "rccl"
    comm = gd.default_comm(ex)
    assert type(comm) is gd.TorchComm, type(comm)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    comm.all_reduce_sum_(t)
    assert float(t) == world * (world + 1) / 2
    if not mode.endswith("cpu"):
        chk = gd.comm_self_check(ex, comm, n_elems=1000, reps=2)
        assert chk["communicator"] == "TorchComm" and chk["ranks"] == world
    dist.barrier()
    if rank == 0:
        print(f"dist_worker OK mode={mode} world={world}")
    dist.destroy_process_group()


def main():
    mode, grid = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if mode.startswith("fallback"):
        return fallback_case(mode)
    from oracle import gko_oracle as o
    import ginkgo_amd.distributed as gd

    rp, ci, v = o.stencil_csr(3, grid)
    n = grid ** 3
    part = gd.Partition.build_slabs(grid, world)
    lo, hi = part.range_of(rank)
    comm = gd.TorchComm()
    if mode == "cpu":
        from cpu_backend import CpuCsr, OracleBackend
        be = OracleBackend()
        owned = CpuCsr(hi - lo, n, (rp[lo:hi + 1] - rp[lo]).astype(np.int32), ci[rp[lo]:rp[hi]],
                       v[rp[lo]:rp[hi]])
    else:
        import ginkgo_amd as g
        ex = g.Cdna4Executor.create(0)
        be = gd.HipBackend(ex)
        if mode == "gpu-ipc":
            # the library's own transport between the processes sharing cuda:0: everything the
            # device-resident path does at N > 1 (forks, side stream, gated product) runs for real
            comm = gd.IpcComm(ex)
        planes = gd.Partition.build_from_global_size_uniform(world, grid).offsets
        owned = g.stencil_csr(ex, 3, grid, z0=planes[rank], nz=planes[rank + 1] - planes[rank])
        assert np.array_equal(owned.col_idxs.cpu().numpy(), ci[rp[lo]:rp[hi]])
    a = gd.DistributedMatrix(be, comm, part, owned)
    # --- halo plan sanity: interior ranks exchange two planes, edge ranks one
    expect = grid * grid * ((rank > 0) + (rank < world - 1))
    assert a.n_halo == expect and a.n_send == expect, (a.n_halo, a.n_send, expect)
    # --- SpMV vs the single-domain oracle
    xg = np.random.default_rng(42).uniform(-1, 1, n)
    x = be.vector_from(xg[lo:hi])
    y = be.vector(hi - lo)
    a.apply(x, y)
    ref = o.csr_spmv(rp, ci, v, xg)[lo:hi]
    got = y.to_numpy()[:, 0]
    err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
    assert err < 1e-14, f"rank {rank}: spmv err {err}"
    # rows without non-local entries keep the exact single-domain summation
    interior = slice(grid * grid if rank > 0 else 0, (hi - lo) - (grid * grid if rank < world - 1 else 0))
    assert np.array_equal(got[interior], ref[interior])
    if mode != "cpu" and "full" in a.nl:
        # slab fast path of the HIP backend: the boundary rows are computed as COMPLETE rows over
        # [x | halo] in the original column order - the whole distributed product has the bits of
        # the single-domain one
        assert a.use_full_boundary and np.array_equal(got, ref)
    # --- CG + block-Jacobi(8) vs the single-process oracle solve
    solver = gd.DistributedCg(be, comm, a, 500, 1e-10, 8)
    xs = be.vector(hi - lo)
    solver.apply(be.vector_from(np.ones(hi - lo)), xs)
    xo, iters, _ = o.cg_solve(rp, ci, v, np.ones(n), max_iters=500, reduction=1e-10, precond="block")
    assert abs(solver.num_iterations - iters) <= 1, (solver.num_iterations, iters)
    e = np.linalg.norm(xs.to_numpy()[:, 0] - xo[lo:hi]) / np.linalg.norm(xo[lo:hi])
    assert e < 1e-8, f"rank {rank}: cg err {e}"
    # --- the asynchronous criterion check (masked run-ahead) must not change anything
    solver0 = gd.DistributedCg(be, comm, a, 500, 1e-10, 8, check_lag=0)
    xs0 = be.vector(hi - lo)
    solver0.apply(be.vector_from(np.ones(hi - lo)), xs0)
    assert solver.check_lag > 0
    assert solver0.num_iterations == solver.num_iterations
    assert np.array_equal(xs0.to_numpy(), xs.to_numpy())
    # --- distributed PipeCg (one all-reduce per iteration) vs the single-process oracle PipeCg
    pipe = gd.DistributedPipeCg(be, comm, a, 500, 1e-10, 8)
    xp = be.vector(hi - lo)
    pipe.apply(be.vector_from(np.ones(hi - lo)), xp)
    xo_p, it_p, _ = o.krylov_solve("pipe_cg", rp, ci, v, np.ones(n), max_iters=500, reduction=1e-10,
                                   precond="block")
    assert abs(pipe.num_iterations - it_p) <= 1, (pipe.num_iterations, it_p)
    e = np.linalg.norm(xp.to_numpy()[:, 0] - xo_p[lo:hi]) / np.linalg.norm(xo_p[lo:hi])
    assert e < 1e-8, f"rank {rank}: pipe_cg err {e}"
    pipe0 = gd.DistributedPipeCg(be, comm, a, 500, 1e-10, 8, check_lag=0, fused=False, taped=False)
    xp0 = be.vector(hi - lo)
    pipe0.apply(be.vector_from(np.ones(hi - lo)), xp0)
    assert pipe0.num_iterations == pipe.num_iterations, (pipe0.num_iterations, pipe.num_iterations)
    if mode == "cpu":
        # same kernels, only the run-ahead differs: identical bits
        assert np.array_equal(xp0.to_numpy(), xp.to_numpy())
    else:
        # fused step_1 + dots uses another reduction tree than the three separate reductions
        d = np.linalg.norm(xp0.to_numpy() - xp.to_numpy()) / np.linalg.norm(xp.to_numpy())
        assert d < 1e-9, d
    # --- fused producer+reduction kernels (HipBackend only) vs the plain sequence
    if mode != "cpu":
        solver1 = gd.DistributedCg(be, comm, a, 500, 1e-10, 8, fused=False)
        xs1 = be.vector(hi - lo)
        solver1.apply(be.vector_from(np.ones(hi - lo)), xs1)
        assert abs(solver1.num_iterations - solver.num_iterations) <= 1
        d = np.linalg.norm(xs1.to_numpy() - xs.to_numpy()) / np.linalg.norm(xs.to_numpy())
        assert d < 1e-9, d
    # --- <p,q> fused into the local SpMV + the boundary rows' share next to their update (the
    #     default only for large local parts): same iterations, same solution to rounding
    if mode != "cpu":
        a.fused_dot_min_rows = 0
        solver2 = gd.DistributedCg(be, comm, a, 500, 1e-10, 8)
        xs2 = be.vector(hi - lo)
        solver2.apply(be.vector_from(np.ones(hi - lo)), xs2)
        a.fused_dot_min_rows = 1 << 22
        assert abs(solver2.num_iterations - solver.num_iterations) <= 1
        d = np.linalg.norm(xs2.to_numpy() - xs.to_numpy()) / np.linalg.norm(xs.to_numpy())
        assert d < 1e-9, d
    # --- restarted GMRES on the distributed matrix (HIP kernels only)
    if mode != "cpu":
        # (a restart length that converges within the iteration limit also on the 64^3 grid of the
        # 8-rank test: GMRES(10) stalls there)
        kd = 10 if grid <= 32 else 40
        for ortho in ("mgs", "cgs"):
            gm = gd.DistributedGmres(be, comm, a, 400, 1e-9, 8, krylov_dim=kd, ortho_method=ortho)
            xg_ = be.vector(hi - lo)
            gm.apply(be.vector_from(np.ones(hi - lo)), xg_)
            xo_, it_, _ = o.gmres_solve(rp, ci, v, np.ones(n), krylov_dim=kd, ortho=ortho, max_iters=400,
                                        reduction=1e-9, precond="block", max_block_size=8)
            assert gm.has_converged and abs(gm.num_iterations - it_) <= 1, (gm.num_iterations, it_)
            e = np.linalg.norm(xg_.to_numpy()[:, 0] - xo_[lo:hi]) / np.linalg.norm(xo_[lo:hi])
            assert e < 1e-7, f"rank {rank}: gmres({ortho}) err {e}"
    # --- a general row partition: irregular symmetric positive definite matrix whose
    #     rows reach into EVERY other rank (halo from several peers, scattered
    #     indices, uneven part sizes) - nothing slab-specific may be assumed
    irregular_case(mode, o, gd, be, comm, rank, world)
    flan_case(mode, o, gd, be, comm, rank, world)
    heavy_tail_case(mode, o, gd, be, comm, rank, world)
    if mode == "gpu-ipc":
        if grid % 8 == 0:
            assert a._gate is not None, "the one-kernel gated product was not taken on the device-resident transport"
        comm.check()
        comm.close()
    dist.barrier()
    if rank == 0:
        print(f"dist_worker OK mode={mode} world={world} grid={grid} iters={solver.num_iterations}")
    dist.destroy_process_group()


def irregular_case(mode, o, gd, be, comm, rank, world):
    import scipy.sparse as sp
    n = 613
    rng = np.random.default_rng(77)           # same matrix on every rank
    m = sp.random(n, n, density=0.02, random_state=rng, format="csr",
                  data_rvs=lambda k: rng.uniform(-1, 1, k))
    m = m + m.T
    m = (m + sp.diags(np.asarray(abs(m).sum(axis=1)).ravel() + 1.0)).tocsr()
    m.sort_indices()
    rp, ci, v = m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data
    part = gd.Partition.build_from_global_size_uniform(world, n)     # 205 / 204 / 204 for 3 ranks
    lo, hi = part.range_of(rank)
    lrp = (rp[lo:hi + 1] - rp[lo]).astype(np.int32)
    lci, lv = ci[rp[lo]:rp[hi]], v[rp[lo]:rp[hi]]
    if mode == "cpu":
        from cpu_backend import CpuCsr
        owned = CpuCsr(hi - lo, n, lrp, lci, lv)
    else:
        import ginkgo_amd as g
        owned = g.Csr.from_arrays(be.exec, (hi - lo, n), lrp, lci, lv)
    a = gd.DistributedMatrix(be, comm, part, owned)
    if world > 2:
        assert sum(1 for c in a.recv_counts if c > 0) >= 2, a.recv_counts   # several peers
    xg = rng.uniform(-1, 1, n)
    x, y = be.vector_from(xg[lo:hi]), be.vector(hi - lo)
    a.apply(x, y)
    ref = o.csr_spmv(rp, ci, v, xg)[lo:hi]
    err = np.max(np.abs(y.to_numpy()[:, 0] - ref)) / np.max(np.abs(ref))
    assert err < 1e-14, f"rank {rank}: irregular spmv err {err}"
    solver = gd.DistributedCg(be, comm, a, 300, 1e-10, 0)          # unpreconditioned
    xs = be.vector(hi - lo)
    rhs = np.sin(np.arange(n))
    solver.apply(be.vector_from(rhs[lo:hi]), xs)
    xo, iters, _ = o.cg_solve(rp, ci, v, rhs, max_iters=300, reduction=1e-10)
    assert abs(solver.num_iterations - iters) <= 1, (solver.num_iterations, iters)
    e = np.linalg.norm(xs.to_numpy()[:, 0] - xo[lo:hi]) / np.linalg.norm(xo[lo:hi])
    assert e < 1e-8, f"rank {rank}: irregular cg err {e}"


def flan_case(mode, o, gd, be, comm, rank, world):
    """configs[4]'s stand-in (ginkgo_amd/workloads.py: L27(g) (x) B3, 3 unknowns per node, up to 81
    entries per row) on an ENTRY-balanced contiguous partition: the product in CSR and - on the GPU -
    with the local block in SELL-P, CG + block-Jacobi(3), against the single-domain oracle"""
    from ginkgo_amd import workloads as wl
    grid = 7
    n, nnz = wl.flan_like_dims(grid)
    rp, ci, v = wl.flan_like_rows(grid)
    assert rp[-1] == nnz
    offsets = wl.partition_by_nnz(wl.flan_like_row_prefix(grid), world, align=3)
    part = gd.Partition(offsets)
    lo, hi = part.range_of(rank)
    lrp, lci, lv = wl.flan_like_rows(grid, lo, hi)
    xg = np.random.default_rng(11).uniform(-1, 1, n)
    ref = o.csr_spmv(rp, ci, v, xg)[lo:hi]
    formats = ("csr",) if mode == "cpu" else ("csr", "sellp")
    for fmt in formats:
        if mode == "cpu":
            from cpu_backend import CpuCsr
            owned = CpuCsr(hi - lo, n, lrp, lci, lv)
            a = gd.DistributedMatrix(be, comm, part, owned)
        else:
            import ginkgo_amd as g
            owned = g.Csr.from_arrays(be.exec, (hi - lo, n), lrp, lci, lv)
            a = gd.DistributedMatrix(be, comm, part, owned, local_format=fmt)
        x, y = be.vector_from(xg[lo:hi]), be.vector(hi - lo)
        a.apply(x, y)
        err = np.max(np.abs(y.to_numpy()[:, 0] - ref)) / np.max(np.abs(ref)) if hi > lo else 0.0
        assert err < 1e-14, f"rank {rank}: flan-like spmv ({fmt}) err {err}"
        solver = gd.DistributedCg(be, comm, a, 400, 1e-10, 3)
        xs = be.vector(hi - lo)
        solver.apply(be.vector_from(np.ones(hi - lo)), xs)
        xo, iters, _ = o.cg_solve(rp, ci, v, np.ones(n), max_iters=400, reduction=1e-10, precond="block",
                                  max_block_size=3)
        assert abs(solver.num_iterations - iters) <= 1, (fmt, solver.num_iterations, iters)
        if hi > lo:
            e = np.linalg.norm(xs.to_numpy()[:, 0] - xo[lo:hi]) / np.linalg.norm(xo[lo:hi])
            assert e < 1e-8, f"rank {rank}: flan-like cg ({fmt}) err {e}"


def heavy_tail_case(mode, o, gd, be, comm, rank, world):
    """configs[4]'s irregular stand-in (workloads.irregular_rows: power-law row lengths, hub rows linked to
    rows of EVERY rank) on an entry-balanced contiguous partition: each rank builds its own rows only; the
    distributed product in CSR and - on the GPU - SELL-P, CG + block-Jacobi(4), against the single-domain
    oracle (core/distributed/matrix.cpp:450-509, partition.hpp:262)"""
    from ginkgo_amd import workloads as wl
    n = 6000
    rp, ci, v = wl.irregular_rows(n)
    prefix = wl.irregular_row_prefix(n)
    assert np.array_equal(prefix, rp.astype(np.int64))
    offsets = wl.partition_by_nnz(prefix, world, align=4)
    part = gd.Partition(offsets)
    lo, hi = part.range_of(rank)
    lrp, lci, lv = wl.irregular_rows(n, lo, hi)
    assert np.array_equal(lci, ci[rp[lo]:rp[hi]]) and np.array_equal(lv, v[rp[lo]:rp[hi]])
    shares = np.diff(prefix[offsets])
    assert shares.max() <= 1.25 * prefix[-1] / world + 400           # (a hub row cannot be split)
    xg = np.random.default_rng(13).uniform(-1, 1, n)
    ref = o.csr_spmv(rp, ci, v, xg)[lo:hi]
    xo, iters, _ = o.cg_solve(rp, ci, v, np.ones(n), max_iters=400, reduction=1e-10, precond="block",
                              max_block_size=4)
    for fmt in (("csr",) if mode == "cpu" else ("csr", "sellp")):
        if mode == "cpu":
            from cpu_backend import CpuCsr
            a = gd.DistributedMatrix(be, comm, part, CpuCsr(hi - lo, n, lrp, lci, lv))
        else:
            import ginkgo_amd as g
            owned = g.Csr.from_arrays(be.exec, (hi - lo, n), lrp, lci, lv)
            a = gd.DistributedMatrix(be, comm, part, owned, local_format=fmt)
        if world > 1:
            assert sum(1 for c in a.recv_counts if c > 0) == world - 1, "the hubs reach every rank"
        x, y = be.vector_from(xg[lo:hi]), be.vector(hi - lo)
        a.apply(x, y)
        err = np.max(np.abs(y.to_numpy()[:, 0] - ref)) / np.max(np.abs(ref))
        assert err < 1e-14, f"rank {rank}: heavy-tailed spmv ({fmt}) err {err}"
        solver = gd.DistributedCg(be, comm, a, 400, 1e-10, 4)
        xs = be.vector(hi - lo)
        solver.apply(be.vector_from(np.ones(hi - lo)), xs)
        assert abs(solver.num_iterations - iters) <= 1, (fmt, solver.num_iterations, iters)
        e = np.linalg.norm(xs.to_numpy()[:, 0] - xo[lo:hi]) / np.linalg.norm(xo[lo:hi])
        assert e < 1e-8, f"rank {rank}: heavy-tailed cg ({fmt}) err {e}"


if __name__ == "__main__":
    main()
