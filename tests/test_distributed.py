"""Multi-rank path: world_size-2/3 gloo runs on CPU of the
partition / halo-plan / exchange / all-reduce logic (oracle-backed kernels), and
on the GPU box a 2-rank run with the HIP kernels (both ranks on cuda:0, exchange
staged through gloo).  Mirrors test/mpi/distributed/{matrix,vector}.cpp and
test/mpi/solver/solver.cpp (ranks launched on one node)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from util import record_perf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(mode, world, grid, timeout=600, extra_env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), mode, str(grid)]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    if world > 2:
        # several ranks share ONE GPU here: each one's memory-class survey is kept short (they also take
        # turns, csrc/arena.hip walk_turn) - the layout of the arena is not what these tests are about
        env.setdefault("GKOC_ARENA_MAX_WALK", "24")
    for attempt in range(2):       # one retry: the rendezvous port can race
        cmd[cmd.index("--master-port") + 1] = str(_free_port())
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        if p.returncode == 0:
            break
    assert p.returncode == 0, p.stdout[-3000:] + "\n--- stderr ---\n" + p.stderr[-12000:]
    assert "dist_worker OK" in p.stdout


def test_partition_helpers():
    import ginkgo_amd.distributed as gd
    p = gd.Partition.build_from_global_size_uniform(3, 10)
    assert p.offsets == [0, 4, 7, 10]          # partition.hpp:262 semantics
    assert p.owner_of(np.array([0, 3, 4, 6, 7, 9])).tolist() == [0, 0, 1, 1, 2, 2]
    s = gd.Partition.build_slabs(8, 3)
    assert s.offsets == [0, 3 * 64, 6 * 64, 8 * 64]
    with pytest.raises(Exception):
        gd.Partition([0, 5, 3])


# (8, 8): one plane per rank - every row of an interior rank is a boundary row with halo from
# both sides; (8, 16): the 8-rank slab partition of the bench; (5, 12): uneven slabs (3,3,2,2,2)
@pytest.mark.parametrize("world,grid", [(2, 8), (3, 9), (8, 8), (8, 16), (5, 12)])
def test_distributed_cpu_gloo(world, grid):
    _launch("cpu", world, grid)


@pytest.mark.parametrize("stage", ["load", "init"])
def test_default_comm_fallback_is_taken_by_all_ranks_cpu(stage):
    """a failure of the RCCL bring-up injected on rank 1 of 3: all ranks fall back together"""
    _launch("fallback-cpu", 3, 0, extra_env={"GKO_COMM_INJECT_FAIL": f"1:{stage}"})


@pytest.mark.gpu
def test_distributed_gpu_two_ranks_one_device():
    _launch("gpu", 2, 16)


@pytest.mark.gpu
def test_distributed_gpu_three_ranks_odd_planes():
    """3 ranks, 9^3: planes of 81 rows - the interior-rows view of the local block starts at a row
    pointer that is only 4-byte aligned"""
    _launch("gpu", 3, 9)


@pytest.mark.gpu
@pytest.mark.parametrize("grid", [32] + ([64] if os.environ.get("GKO_TEST_FULL_SOLVE") == "1" else []))
def test_distributed_gpu_eight_ranks_one_device(grid):
    """world_size = 8 (BASELINE configs[3]'s rank count): eight processes sharing cuda:0,
    HIP kernels, exchange staged through gloo.  32^3 = 4-plane slabs, 64^3 = 8-plane slabs;
    DistributedCg / PipeCg / Gmres against the single-domain oracle (iterations +-1, solution
    1e-8, boundary rows 1e-14, interior rows bit-exact) - tests/dist_worker.py"""
    _launch("gpu", 8, grid, timeout=1500)


@pytest.mark.gpu
def test_default_comm_fallback_is_taken_by_all_ranks_gpu():
    """the same injection with the real executor (2 ranks on cuda:0), followed by the
    communicator self-check of bench.py on the fallback communicator"""
    _launch("fallback-gpu", 2, 0, extra_env={"GKO_COMM_INJECT_FAIL": "1:load"})


def _bench(world, args, env_extra, timeout=1500):
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world)] + args
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    if world > 2 and env.get("GKO_BENCH_BACKEND") == "gloo":
        env.setdefault("GKOC_ARENA_MAX_WALK", "24")      # ranks sharing one GPU: short surveys (see _launch)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + "\n--- stderr ---\n" + p.stderr[-12000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-3000:]          # rank 0 only, one line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_command_path_with_eight_ranks():
    """the driver's exact multi-GPU command (torch.distributed.run ... bench.py --gpus 8 --steps K
    --warmup W) on one GPU: GKO_BENCH_BACKEND=gloo lets the 8 ranks share cuda:0.  Process group,
    SlabPartition(.., 8), communicator self-check, max-over-ranks timing, JSON from rank 0 only."""
    G = 32                      # 4-plane slabs: the suite's budget; 64 ran in rounds 2-4
    d = _bench(8, ["--steps", "3", "--warmup", "1", "--grid", str(G), "--cg-iters", "20", "--pipe-cg"],
               {"GKO_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["config"]["partition"] == "8 z-slab(s)"
    n, nnz = G ** 3, (3 * G - 2) ** 3
    assert f"n={n}, nnz={nnz}" in d["config"]["workload"]
    assert d["cg_iterations"] == 20 and d["pipe_cg_iterations"] == 20 and d["cg_iters_per_s"] > 0
    assert d["comm_check"]["ranks"] == 8 and d["comm_check"]["exchange_us"] > 0
    prof = d["rank0_profile"]
    assert prof["n_local_rows"] == n // 8 and prof["n_halo"] == G * G      # rank 0: one neighbour
    assert prof["local_spmv_ms"] > 0


@pytest.mark.gpu
def test_bench_command_path_on_the_device_resident_transport():
    """the same command with GKO_COMM=ipc: the eight ranks sharing cuda:0 talk through the library's
    mailbox transport, so bench.py's N > 1 path is the DEVICE-RESIDENT one the 8-GPU run takes (forks,
    side stream, one-kernel gated product, PipeCg with gated steps) - not the host-staged gloo one"""
    d = _bench(8, ["--steps", "3", "--warmup", "1", "--grid", "32", "--cg-iters", "20", "--pipe-cg"],
               {"GKO_BENCH_BACKEND": "gloo", "GKO_COMM": "ipc", "GKOC_IPC_PATIENCE_MS": "60000"})
    assert d["n_gpus"] == 8 and d["cg_iterations"] == 20 and d["pipe_cg_iterations"] == 20
    assert d["comm_check"]["communicator"] == "IpcComm" and d["comm_check"]["transport_choice"]["chosen"] == "IpcComm"
    assert d["distributed_product"]["one_kernel_product"], d["distributed_product"]
    record_perf("bench_eight_ranks_one_gpu_mailbox", comm_check=d["comm_check"], cg_iters_per_s=d["cg_iters_per_s"])


# ---- first contact with a real multi-GPU node must be survivable: every fault below
# ---- has to end in ONE printed line that says what ran - never a hang, never a traceback without a line
_FAULT_ARGS = ["--steps", "2", "--warmup", "1", "--grid", "32", "--no-pipe-cg"]


@pytest.mark.gpu
def test_bench_survives_a_denied_ipc_open_on_one_rank():
    """hipIpcOpenMemHandle refused on rank 1 (GKOC_IPC_INJECT_OPEN_FAIL): the mailbox transport is given up by
    ALL ranks together, the data path goes through the process group, the line says so"""
    d = _bench(4, [*_FAULT_ARGS, "--cg-iters", "10"],
               {"GKO_BENCH_BACKEND": "gloo", "GKO_COMM": "ipc", "GKOC_IPC_INJECT_OPEN_FAIL": "1"}, timeout=600)
    assert d["comm_check"]["transport_choice"]["chosen"] == "TorchComm", d["comm_check"]
    assert d["comm_check"]["communicator"] == "TorchComm" and d["cg_iterations"] == 10
    assert d["distributed_product"]["one_kernel_product"] is False


@pytest.mark.gpu
def test_bench_survives_an_all_reduce_that_runs_out_of_patience():
    """rank 1 stops contributing to the all-reduces in the middle of the warm-up solve (GKOC_IPC_INJECT_MUTE): its
    peers' waits run out of patience ONCE (3 s here), every later wait gives up within a millisecond, the solve
    ends, all ranks agree that the transport is dead and measure on the process group instead"""
    d = _bench(4, [*_FAULT_ARGS, "--cg-iters", "100"],
               {"GKO_BENCH_BACKEND": "gloo", "GKO_COMM": "ipc", "GKOC_IPC_INJECT_MUTE": "1:150",
                "GKOC_IPC_PATIENCE_MS": "3000"}, timeout=600)
    tc = d["comm_check"]["transport_choice"]
    assert tc["chosen"] == "TorchComm" and tc["first_choice"]["chosen"] == "IpcComm", tc
    assert "failed after it had come up" in tc["why"] and d["cg_iterations"] == 100


@pytest.mark.gpu
def test_bench_survives_a_failed_rccl_init_on_one_rank():
    """ncclCommInitRank fails on rank 1 (GKO_COMM_INJECT_FAIL=1:init): no rank is left inside a collective,
    the line names the communicator that carried the data"""
    d = _bench(2, [*_FAULT_ARGS, "--cg-iters", "10"],
               {"GKO_BENCH_BACKEND": "gloo", "GKO_COMM": "rccl", "GKO_COMM_INJECT_FAIL": "1:load"}, timeout=600)
    assert d["comm_check"]["transport_choice"]["chosen"] == "TorchComm", d["comm_check"]
    assert d["cg_iterations"] == 10


@pytest.mark.gpu
def test_bench_line_when_one_rank_finds_a_single_memory_class():
    """the allocator of ONE rank is held to one class (the others survey as usual): the line is printed, the
    per-rank part shows which rank it was, and the minimum over ranks is there to read"""
    d = _bench(4, [*_FAULT_ARGS, "--cg-iters", "10"],
               {"GKO_BENCH_BACKEND": "gloo", "GKO_COMM": "ipc", "GKO_TEST_ONE_CLASS_RANK": "2"}, timeout=600)
    pr = d["roofline"]["per_rank"]
    assert pr[2]["memory_classes_found"] <= 1 and d["roofline"]["memory_classes_min_over_ranks"] <= 1
    assert "search_ms_max_over_ranks" in d["roofline"] and d["cg_iterations"] == 10


@pytest.mark.gpu
def test_peers_on_other_devices_plain_windows_are_refused_and_the_gate_pays_the_full_fence():
    """what a multi-GPU node changes, exercised on one GPU by letting rank 1 REPORT another PCI bus id
    (GKOC_COMM_FAKE_BUS_ID): (a) windows in plain device memory are refused by every rank alike -> no mailbox
    transport; (b) with uncached windows the transport comes up, the gated kernels start with the system-scope
    fence and the cheap gate is trusted only after the product's soak on that communicator has passed"""
    fake = {"GKO_BENCH_BACKEND": "gloo", "GKO_COMM": "ipc", "GKOC_COMM_FAKE_BUS_ID": "1=ffff:ff:1f.7"}
    d = _bench(2, [*_FAULT_ARGS, "--cg-iters", "10"], dict(fake, GKOC_IPC_WINDOW="plain"), timeout=600)
    assert d["comm_check"]["transport_choice"]["chosen"] == "TorchComm", d["comm_check"]
    d = _bench(2, [*_FAULT_ARGS, "--cg-iters", "10"], fake, timeout=600)
    topo = d["comm_check"]["topology"]
    assert d["comm_check"]["communicator"] == "IpcComm" and topo["cross_device"] and topo["window_uncached"]
    assert topo["ranks_seen"] == 2 and topo["bus_ids"][1] == "ffff:ff:1f.7" and topo["bus_ids"][0] != topo["bus_ids"][1]
    dp = d["distributed_product"]
    assert dp["peers_on_other_devices"] and dp["one_kernel_product"]
    assert "trusted after the soak" in dp["gate_fence"] and "64 rounds" in dp["self_check"], dp
    d = _bench(2, [*_FAULT_ARGS, "--cg-iters", "10"], dict(fake, GKO_GATE_TRUST="0"), timeout=600)
    assert "system-scope" in d["distributed_product"]["gate_fence"], d["distributed_product"]


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_and_the_line_has_every_key():
    """`python bench.py --gpus 8 ...` WITHOUT a launcher: the script starts
    its ranks itself; the N > 1 line carries cpu_baseline (the N = 1 figure, labelled), the per-rank
    roofline, the communicator check and what every rank's allocator found."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--grid", "32", "--steps", "3",
           "--warmup", "1", "--cg-iters", "10"]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", GKO_BENCH_BACKEND="gloo")
    env.setdefault("GKOC_ARENA_MAX_WALK", "24")          # eight ranks on one GPU: short surveys (see _launch)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + "\n--- stderr ---\n" + p.stderr[-12000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["partition"] == "8 z-slab(s)"
    assert "cpu_baseline" in d and "no CPU twin at N > 1" in d["cpu_baseline"]["note"]
    pr = d["roofline"]["per_rank"]
    assert len(pr) == 8 and all(r["kernel_ms"] > 0 and r["achieved"] is not None for r in pr)
    # (0: a rank whose survey found nothing it could use next to seven others on the same device works
    # with one hipMalloc per array - what the allocator found is recorded in the line, not asserted)
    assert all(0 <= r["memory_classes_found"] <= 3 for r in pr), [r["memory_classes_found"] for r in pr]
    assert d["comm_check"]["ranks"] == 8 and "memory_classes_found" in d["config"]
    assert d["cg_iterations"] == 10 and d["pipe_cg_iterations"] == 10


@pytest.mark.gpu
def test_bench_configs4_irregular_stand_in_one_and_eight_ranks():
    """`bench.py --workload irregular`: the heavy-tailed stand-in - power-law row
    lengths, hub rows beyond GKOC_CSR_LONG_ROW that reach into every rank - SELL-P vs CSR on one GPU and on 8
    ranks (entry-balanced contiguous rows); the line says what the matrix looks like"""
    common = ["--workload", "irregular", "--irr-n", "120000", "--steps", "3", "--warmup", "1", "--cg-iters", "10",
              "--block-size", "4"]
    d1 = _bench_plain(["--format", "sellp", *common])
    w = d1["config"]["workload"]
    assert "irregular stand-in" in w and "n = 120000" in w and "rows beyond GKOC_CSR_LONG_ROW" in w
    assert d1["formats"]["sellp"]["stored_over_nnz"] > 1.3 and d1["formats"]["csr"]["stored_over_nnz"] == 1.0
    assert d1["formats"]["csr"]["ms"] > 0 and d1["formats"]["sellp"]["ms"] > 0
    assert d1["cg"]["csr"]["cg_iterations"] == 10 and d1["cg"]["sellp"]["cg_iterations"] == 10
    record_perf("bench_irregular_stand_in_120k", formats=d1["formats"], cg=d1["cg"])
    d8 = _bench_plain(["--gpus", "8", "--format", "csr", *common], {"GKO_BENCH_BACKEND": "gloo"})
    pr = d8["roofline"]["per_rank"]
    assert d8["n_gpus"] == 8 and len(pr) == 8 and sum(r["rows"] for r in pr) == 120000
    assert max(r["nnz"] for r in pr) <= 1.3 * sum(r["nnz"] for r in pr) / 8
    assert d8["formats"]["csr"]["peers"] == 7                     # the hubs reach every rank
    assert d8["cg"]["csr"]["cg_iterations"] == 10 and d8["cg"]["sellp"]["cg_iterations"] == 10


def _bench_plain(args, env_extra=None):
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *args]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + "\n--- stderr ---\n" + p.stderr[-12000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-3000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_configs4_stand_in_one_and_eight_ranks(tmp_path):
    """BASELINE configs[4] as a configuration of bench.py: the Flan_1565
    stand-in and a MatrixMarket file, SELL-P vs CSR, CG + block-Jacobi(3), on one rank and on eight
    (gloo, sharing the GPU); the eight-rank numerics against the oracle are tests/dist_worker.py's
    flan_case (test_distributed_gpu_eight_ranks_one_device)."""
    common = ["--steps", "3", "--warmup", "1", "--cg-iters", "10", "--flan-grid", "12"]
    d1 = _bench_plain(["--workload", "flan", "--format", "sellp", *common])
    n, nnz = 3 * 12 ** 3, 9 * 34 ** 3
    assert d1["data"] == "synthetic stand-in for Flan_1565" and f"n = {n}, nnz = {nnz}" in d1["config"]["workload"]
    assert d1["config"]["format"] == "sellp" and d1["n_gpus"] == 1
    assert d1["formats"]["sellp"]["bit_identical_to_csr"] and d1["formats"]["sellp"]["stored_over_nnz"] >= 1.0
    assert d1["cg"]["csr"]["cg_iterations"] == 10 and d1["cg"]["sellp"]["cg_iterations"] == 10
    assert d1["roofline"]["frac"] > 0 and d1["cpu_baseline"]["note"]
    d8 = _bench_plain(["--gpus", "8", "--workload", "flan", "--format", "csr", *common],
                      {"GKO_BENCH_BACKEND": "gloo"})
    assert d8["n_gpus"] == 8 and d8["config"]["partition"].startswith("contiguous rows")
    pr = d8["roofline"]["per_rank"]
    assert len(pr) == 8 and sum(r["rows"] for r in pr) == n and sum(r["nnz"] for r in pr) == nnz
    assert max(r["nnz"] for r in pr) <= 1.15 * nnz / 8 and all(r["rows"] % 3 == 0 for r in pr)
    assert d8["cg"]["csr"]["cg_iterations"] == 10 and d8["cg"]["sellp"]["cg_iterations"] == 10
    assert d8["formats"]["csr"]["peers"] >= 1 and d8["comm_check"]["ranks"] == 8
    # ... and from a file: a symmetric MatrixMarket file of the same matrix at a smaller size
    import scipy.sparse as sp
    from ginkgo_amd import workloads as wl
    rp, ci, v = wl.flan_like_rows(6)
    a = sp.csr_matrix((v, ci, rp), shape=(3 * 216, 3 * 216))
    path = str(tmp_path / "flan6.mtx")
    wl.write_mtx(path, a, symmetric=True)
    df = _bench_plain(["--matrix", path, "--format", "csr", "--steps", "3", "--warmup", "1", "--cg-iters", "5"])
    assert df["data"] == "file flan6.mtx" and f"n = {3 * 216}, nnz = {a.nnz}" in df["config"]["workload"]
    assert df["formats"]["sellp"]["bit_identical_to_csr"] and df["cg"]["csr"]["cg_iterations"] == 5


@pytest.mark.gpu
def test_distributed_single_rank_matches_plain(gexec, oracle):
    """world = 1: the distributed wrapper degenerates to the plain SpMV / CG"""
    import torch.distributed as dist
    import ginkgo_amd as g
    import ginkgo_amd.distributed as gd
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}",
                                rank=0, world_size=1)
    grid = 12
    part = gd.SlabPartition(grid, 1)
    op = gd.DistributedStencil(gexec, part, 0)
    rp, ci, v = oracle.stencil_csr(3, grid)
    assert op.global_nnz == len(v) and op.matrix.n_halo == 0
    x = op.random_vector(42)
    y = op.zeros_vector()
    op.apply(x, y)
    xg = np.random.default_rng(42).uniform(-1, 1, grid ** 3)
    assert np.array_equal(y.to_numpy()[:, 0], oracle.csr_spmv(rp, ci, v, xg))
    op.prepare_cg(5, lambda: gexec.synchronize())
    iters, t = op.timed_cg(lambda: gexec.synchronize())
    assert iters == 5
    dist.destroy_process_group()


@pytest.mark.gpu
def test_overlap_branch_with_mirror_comm(gexec, oracle):
    """The device-resident exchange branch of DistributedMatrix.apply (halo exchange
    on a second stream, overlapped with the local SpMV, event-ordered) is what runs
    under RCCL.  On one GPU it is exercised with a communicator that plays the other
    rank of a 2-slab run of a z-mirror-symmetric problem: by symmetry the plane rank 1
    would send is the plane rank 0 sends, so "exchange" = device copy on the side
    stream and "all-reduce" = times two.  Checked against the single-domain oracle."""
    import torch
    import ginkgo_amd as g
    import ginkgo_amd.distributed as gd

    grid = 16
    plane, n = grid * grid, grid ** 3

    class MirrorComm:
        rank, size, host_staging = 0, 2, False

        def all_reduce_sum_(self, t):
            return t.mul_(2)

        def all_to_all_counts(self, send_counts):
            return list(send_counts)

        def all_to_all_v(self, recv, send, recv_counts, send_counts, async_op=False):
            assert list(recv_counts) == [0, plane] and list(send_counts) == [0, plane]
            if recv.dtype == torch.int64:        # set-up: the indices the peer wants from us
                recv.copy_(send - plane)
            else:                                # apply: the peer's boundary plane
                recv.copy_(send)
            return None

    rp, ci, v = oracle.stencil_csr(3, grid)
    part = gd.SlabPartition(grid, 2)
    lo, hi = part.range_of(0)
    owned = g.stencil_csr(gexec, 3, grid, z0=0, nz=grid // 2)
    be = gd.HipBackend(gexec)
    a = gd.DistributedMatrix(be, MirrorComm(), part, owned)
    assert a._side is not None and a.n_halo == plane and a.n_send == plane
    half = np.random.default_rng(9).uniform(-1, 1, n // 2)
    xg = np.concatenate([half, half.reshape(grid // 2, plane)[::-1].reshape(-1)])   # x[z] = x[N-1-z]
    x = be.vector_from(xg[lo:hi])
    y = be.vector(hi - lo)
    for _ in range(3):                           # repeated: buffers are reused across applies
        a.apply(x, y)
    ref = oracle.csr_spmv(rp, ci, v, xg)[lo:hi]
    got = y.to_numpy()[:, 0]
    assert np.max(np.abs(got - ref)) <= 1e-14 * np.max(np.abs(ref))
    assert np.array_equal(got[:hi - lo - plane], ref[:hi - lo - plane])     # rows without halo entries
    # CG + block-Jacobi(8): rhs = ones is symmetric, so is every iterate
    solver = gd.DistributedCg(be, MirrorComm(), a, 500, 1e-10, 8)
    xs = be.vector(hi - lo)
    solver.apply(be.vector_from(np.ones(hi - lo)), xs)
    xo, iters, _ = oracle.cg_solve(rp, ci, v, np.ones(n), max_iters=500, reduction=1e-10, precond="block")
    assert abs(solver.num_iterations - iters) <= 1
    e = np.linalg.norm(xs.to_numpy()[:, 0] - xo[lo:hi]) / np.linalg.norm(xo[lo:hi])
    assert e < 1e-8
    # restarted GMRES through the same communicator
    gm = gd.DistributedGmres(be, MirrorComm(), a, 400, 1e-9, 8, krylov_dim=12, ortho_method="cgs")
    xg_ = be.vector(hi - lo)
    gm.apply(be.vector_from(np.ones(hi - lo)), xg_)
    xo2, it2, _ = oracle.gmres_solve(rp, ci, v, np.ones(n), krylov_dim=12, ortho="cgs", max_iters=400,
                                     reduction=1e-9, precond="block", max_block_size=8)
    assert gm.has_converged and abs(gm.num_iterations - it2) <= 1
    assert np.linalg.norm(xg_.to_numpy()[:, 0] - xo2[lo:hi]) / np.linalg.norm(xo2[lo:hi]) < 1e-7


def _slab(gexec, oracle_, grid, world, rank, nz=None):
    """one rank's slab of the 27-pt grid^3 problem, split on the device"""
    import ginkgo_amd as g
    import ginkgo_amd.distributed as gd
    part = gd.SlabPartition(grid, world)
    lo, hi = part.range_of(rank)
    z0, z1 = part.plane_offsets[rank], part.plane_offsets[rank + 1]
    owned = g.stencil_csr(gexec, 3, grid, z0=z0, nz=z1 - z0)
    be = gd.HipBackend(gexec)
    local, nl, recv_gidx = be.split(owned, lo, hi, grid ** 3)
    return be, local, nl, recv_gidx, lo, hi


@pytest.mark.gpu
@pytest.mark.parametrize("grid,world,rank", [(12, 3, 1), (16, 2, 0), (9, 3, 2), (20, 4, 2), (12, 12, 5)])
def test_one_kernel_product_has_the_single_domain_bits(gexec, oracle, grid, world, rank):
    """gkoc_csr_spmv_gated_* (interior rows from the rank's local block, the boundary rows as
    complete rows over [local columns | halo] on the last waves of the same launch, behind a gate;
    b = the local vector with the halo behind it on a 128-byte boundary): every row equals the
    single-domain product bit for bit - interior ranks (halo on both sides), first and last rank,
    planes that are not multiples of the 64-row segments, a one-plane slab where EVERY row is a
    boundary row - and the gate counts across repeated products."""
    import torch

    n = grid ** 3
    rp, ci, v = oracle.stencil_csr(3, grid)
    be, local, nl, recv_gidx, lo, hi = _slab(gexec, oracle, grid, world, rank)
    f = nl["full"]
    assert f["gated"] and f["halo_base"] % 32 == 0 and f["halo_base"] >= hi - lo
    assert f["head"] + f["tail"] == nl["n"] and "ext" not in nl      # no second copy of the matrix
    xg = np.random.default_rng(grid + rank).uniform(-1, 1, n)
    ref = oracle.csr_spmv(rp, ci, v, xg)[lo:hi]
    store = gexec.zeros((f["halo_base"] + recv_gidx.numel(),), torch.float64)
    store[:hi - lo] = torch.from_numpy(xg[lo:hi]).to(store.device)
    y = be.vector(hi - lo)
    gate = be.gate_new()
    halo = torch.from_numpy(xg[recv_gidx.cpu().numpy().astype(np.int64)]).to(store.device)
    for rep in range(3):
        store[f["halo_base"]:] = halo if rep != 1 else 0.0      # rep 1: a wrong halo must show
        be.gate_open(torch.cuda.current_stream(), gate)   # the "exchange" is done: same stream, in front
        be.spmv_gated(local, nl, store, y, gate)
        got = y.to_numpy()[:, 0]
        if rep == 1 and recv_gidx.numel():
            assert not np.array_equal(got, ref)
        else:
            assert np.array_equal(got, ref), rep
    assert int(gate[0][0].item()) == 3 and int(gate[0][1].item()) == 0 and gate[1].value == 3
    # the same product with <x_local, y> from its waves (one partial sum per wave, one fold launch):
    # y keeps its bits, the dot is a tree sum of the row products
    dot = be.vector(1)
    y.fill(0.0)
    be.gate_open(torch.cuda.current_stream(), gate)
    be.spmv_gated_dot(local, nl, store, y, gate, dot)
    assert np.array_equal(y.to_numpy()[:, 0], ref)
    want = float(np.dot(xg[lo:hi], ref))
    scale = float(np.dot(np.abs(xg[lo:hi]), np.abs(ref)))
    assert abs(float(dot.to_numpy()[0, 0]) - want) <= 1e-13 * scale       # tolerance: reduction order
    # ... and it is the same value every time (no atomics, fixed tree)
    first = dot.to_numpy().copy()
    be.gate_open(torch.cuda.current_stream(), gate)
    be.spmv_gated_dot(local, nl, store, y, gate, dot)
    assert np.array_equal(dot.to_numpy(), first)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 1000, 65536 + 3])
@pytest.mark.parametrize("case", ["goes on", "converges", "had stopped", "zero prev_rho"])
def test_cg_step_1_with_the_criterion_inside(gexec, n, case):
    """gkoc_x_cg_step_1_check_* = gkoc_implicit_residual_norm_* followed by gkoc_cg_step_1_*: the
    same p bit for bit, the same stop status, the same two flags (reference: core/solver/cg.cpp:150-162
    - the criterion between the reductions and step_1; reference/stop/residual_norm_kernels.cpp:27-90)"""
    import ctypes as C
    import torch
    import ginkgo_amd.distributed as gd
    from ginkgo_amd._lib import call
    be = gd.HipBackend(gexec)
    rng = np.random.default_rng(n + len(case))
    p0, z0 = rng.uniform(-1, 1, max(n, 1))[:n], rng.uniform(-1, 1, max(n, 1))[:n]
    tau0 = 4.0
    tau_sq = {"goes on": 1.0, "converges": 1e-30, "had stopped": 1.0, "zero prev_rho": 1.0}[case]
    prev = 0.0 if case == "zero prev_rho" else 0.7
    res = []
    for fused in (False, True):
        p, z = be.vector_from(p0.copy()) if n else be.vector(0), be.vector_from(z0.copy()) if n else be.vector(0)
        rho, prev_rho = be.scalar(1.3), be.scalar(prev)
        tau, orig = be.scalar(tau_sq), be.scalar(tau0)
        flags, stop = be.stop_flags()
        if case == "had stopped":
            stop.fill_(0x40 | 1)
        host = torch.full((2,), 0xFF, dtype=torch.uint8).pin_memory()
        if fused:
            call("gkoc_x_cg_step_1_check_f64", gexec.stream, n, p.values, z.values, rho.values, prev_rho.values,
                 tau.values, orig.values, C.c_double(1e-10), C.c_int(1), C.c_uint8(2), C.c_int(1), stop, host)
        else:
            call("gkoc_implicit_residual_norm_f64", gexec.stream, 1, tau.values, orig.values, C.c_double(1e-10),
                 C.c_uint8(2), C.c_int(1), stop, host, None, None)
            be.cg_step_1(p, z, rho, prev_rho, stop)
        gexec.synchronize()
        res.append((p.to_numpy().copy(), int(stop.cpu()[0]), host.clone().tolist()))
    assert np.array_equal(res[0][0], res[1][0])
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2], (res[0][1:], res[1][1:])
    if case == "converges":
        assert res[1][1] == (0x80 | 0x40 | 2) and res[1][2] == [1, 1]
        assert np.array_equal(res[1][0].ravel(), p0)      # p left alone
    if case == "goes on" and n:
        assert res[1][2] == [0, 0] and not np.array_equal(res[1][0].ravel(), p0)


def _late_gate(gexec, be, local, nl, recv_gidx, x_local, halo_vals, delay_us=5000):
    """the product on the main stream, its halo and gate_open on a side stream BEHIND a kernel that
    sleeps delay_us and an RCCL-sized kernel (64 workgroups x 512 threads, 32 KB of LDS each) that
    has to find room next to the waiting boundary waves; returns (y, elapsed ms, gate words)"""
    import ctypes as C
    import torch
    from ginkgo_amd._lib import call
    f = nl["full"]
    n = x_local.numel()
    store = gexec.zeros((f["halo_base"] + max(recv_gidx.numel(), 1),), torch.float64)
    store[:n] = x_local
    y = be.vector(n)
    gate = be.gate_new()
    side = be.side_stream()
    main = torch.cuda.current_stream()
    # warm both kernels up with an open gate (first launches load code objects)
    be.gate_open(main, gate)
    be.spmv_gated(local, nl, store, y, gate)
    call("gkoc_debug_delay", C.c_void_p(side.cuda_stream), C.c_int64(10), 1, 64, 0)
    torch.cuda.synchronize()
    store[f["halo_base"]:f["halo_base"] + recv_gidx.numel()] = 0.0    # the halo of the "previous iteration"
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sst = C.c_void_p(side.cuda_stream)
    e0.record(main)
    call("gkoc_debug_delay", sst, C.c_int64(delay_us), 1, 64, 0)        # the exchange is late ...
    with torch.cuda.stream(side):
        store[f["halo_base"]:f["halo_base"] + recv_gidx.numel()].copy_(halo_vals)     # ... arrives ...
    call("gkoc_debug_delay", sst, C.c_int64(20), 64, 512, 32768)        # ... through RCCL-sized kernels
    be.gate_open(side, gate)
    be.spmv_gated(local, nl, store, y, gate)                              # main stream: no event, no join
    e1.record(main)
    torch.cuda.synchronize()
    return y, e0.elapsed_time(e1), [int(v) for v in gate[0].tolist()]


@pytest.mark.gpu
@pytest.mark.parametrize("grid,world,rank", [(256, 8, 3), (64, 64, 7)])
def test_one_kernel_product_waits_for_a_late_halo(gexec, oracle, grid, world, rank):
    """The waiting branch: the halo arrives 5 ms AFTER the product was
    launched, behind kernels on another stream that need room on the device while the boundary waves
    spin - (256, 8): the per-rank slab of the 8-GPU run, 2048 waiting waves; (64, 64): a one-plane
    slab, every wave waits.  The product must wait (about 5 ms, not the 10 s of its give-up), take
    the late halo (bit-identical to the single-domain rows) and leave gate[1] == 0."""
    import torch
    n = grid ** 3
    be, local, nl, recv_gidx, lo, hi = _slab(gexec, oracle, grid, world, rank)
    assert nl["full"]["gated"]
    xg = torch.from_numpy(np.random.default_rng(7).uniform(-1, 1, n)).to(gexec.device)
    halo = xg[recv_gidx.long()]
    y, ms, words = _late_gate(gexec, be, local, nl, recv_gidx, xg[lo:hi], halo)
    assert words[1] == 0, "a boundary wave gave up waiting"
    # the product cannot end before the halo that left 5 ms after e0 (the delay kernel counts the 100 MHz
    # constant clock); how long after is the box's business - "not the 10 s of the give-up" is words[1] == 0
    assert ms > 4.5, ms
    record_perf("one_kernel_product_late_halo", grid=grid, world=world, ms=ms)
    # reference: the same rows through the stream-ordered kernels (bit-identical to the oracle's
    # single-domain rows: test_one_kernel_product_has_the_single_domain_bits)
    import ginkgo_amd as g
    f = nl["full"]
    y2 = be.vector(hi - lo)
    xl = g.Dense(gexec, xg[lo:hi].clone().view(-1, 1))
    hv = g.Dense(gexec, halo.clone().view(-1, 1))
    be.spmv_rows(local, f["interior"][0], f["interior"][1], xl, y2)
    be.rowlist_full(nl, xl, hv, y2)
    assert torch.equal(y.values, y2.values)
    if grid <= 64:
        rp, ci, v = oracle.stencil_csr(3, grid)
        assert np.array_equal(y.to_numpy()[:, 0], oracle.csr_spmv(rp, ci, v, xg.cpu().numpy())[lo:hi])


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["goes on", "converges", "late gate"])
def test_pipe_cg_step_kernel_waits_and_judges_the_criterion_itself(gexec, oracle, case):
    """gkoc_x_pipe_cg_steps_jacobi_*(gate): the wait for the all-reduced scalars (the product's gate
    word) and ImplicitResidualNorm inside the step kernel = a join + gkoc_implicit_residual_norm_* +
    the plain step kernel: same ten vectors, same partial sums, same stop status and flags; "late
    gate": the word is set 3 ms AFTER the kernel was launched, from another stream."""
    import ctypes as C
    import torch
    import ginkgo_amd as g
    import ginkgo_amd.distributed as gd
    from ginkgo_amd._lib import call
    grid = 20
    n = grid ** 3
    be = gd.HipBackend(gexec)
    a = g.stencil_csr(gexec, 3, grid)
    m_op = be.jacobi(a, 8)
    rng = np.random.default_rng(3)
    base = [rng.uniform(-1, 1, n) for _ in range(10)]
    res = []
    for gated in (False, True):
        vecs = [be.vector_from(b.copy()) for b in base]
        x, r, z, w, p, q, f, gg, m, nn = vecs
        trip_t, trip = be.scalar_tuple(3)
        trip_t.copy_(torch.tensor([1.3, 0.9, 1e-30 if case == "converges" else 2.0], dtype=torch.float64))
        prev_rho, tau0 = be.scalar(0.7), be.scalar(4.0)
        b_in, b_out = be.scalar(0.4), be.scalar(0.0)
        out3 = gexec.zeros((3,), torch.float64)
        flags, stop = be.stop_flags()
        slot = be.check_slot()
        if gated:
            gate = be.gate_new()
            side = be.side_stream()
            if case == "late gate":
                call("gkoc_debug_delay", C.c_void_p(side.cuda_stream), C.c_int64(3000), 1, 64, 0)
            be.gate_open(side, gate)
            sg = be.step_gate(gate, trip[2], tau0, 1e-10, stop, slot)
            t0 = torch.cuda.Event(enable_timing=True)
            t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            assert be.pipe_cg_steps_jacobi(m_op, x, r, z, w, p, q, f, gg, m, nn, prev_rho, trip[0], trip[1],
                                           b_in, b_out, stop, out3, gate=sg)
            t1.record()
            gexec.synchronize()
            assert int(gate[0][1].item()) == 0
            if case == "late gate":
                assert t0.elapsed_time(t1) > 2.5        # it waited for the 3 ms late gate (gate[0][1] == 0: no give-up)
        else:
            call("gkoc_implicit_residual_norm_f64", gexec.stream, 1, trip[2].values, tau0.values,
                 C.c_double(1e-10), C.c_uint8(2), C.c_int(1), stop, be._chk_host[slot], None, None)
            assert be.pipe_cg_steps_jacobi(m_op, x, r, z, w, p, q, f, gg, m, nn, prev_rho, trip[0], trip[1],
                                           b_in, b_out, stop, out3)
            gexec.synchronize()
        res.append(([v.to_numpy().copy() for v in vecs], out3.cpu().numpy().copy(), int(stop.cpu()[0]),
                    be._chk_host[slot].clone().tolist(), float(b_out.to_numpy()[0, 0])))
    for va, vb in zip(res[0][0], res[1][0]):
        assert np.array_equal(va, vb)
    assert np.array_equal(res[0][1], res[1][1]) and res[0][2:] == res[1][2:], (res[0][2:], res[1][2:])
    if case == "converges":
        assert res[1][2] == (0x80 | 0x40 | 2) and res[1][3] == [1, 1]
        assert np.array_equal(res[1][0][0][:, 0], base[0])          # x left alone
    else:
        assert res[1][3] == [0, 0] and not np.array_equal(res[1][0][0][:, 0], base[0])


@pytest.mark.gpu
@pytest.mark.parametrize("with_dot", [False, True])
def test_product_opens_the_fork_of_its_own_exchange(gexec, oracle, with_dot):
    """gkoc_csr_spmv_gated_*(fork_word, fork_number): the product's first wave stores the number,
    the exchange's stream - enqueued BEFORE the product, polling with gkoc_stream_fork_wait - then
    delivers the halo and opens the gate, the boundary waves of the same launch take it.  No event,
    no kernel in front of the product.  Three products in a row (the numbers count), single-domain
    bits every time."""
    import ctypes as C
    import torch
    from ginkgo_amd._lib import call
    grid, world, rank = 64, 8, 3
    n = grid ** 3
    rp, ci, v = oracle.stencil_csr(3, grid)
    be, local, nl, recv_gidx, lo, hi = _slab(gexec, oracle, grid, world, rank)
    f = nl["full"]
    store = gexec.zeros((f["halo_base"] + recv_gidx.numel(),), torch.float64)
    y, dot = be.vector(hi - lo), be.vector(1)
    gate = be.gate_new()
    side = be.side_stream()
    sst = C.c_void_p(side.cuda_stream)
    word = gexec.zeros((64,), torch.int32)
    for k in range(1, 4):
        xg = np.random.default_rng(k).uniform(-1, 1, n)
        ref = oracle.csr_spmv(rp, ci, v, xg)[lo:hi]
        halo = torch.from_numpy(xg[recv_gidx.cpu().numpy().astype(np.int64)]).to(store.device)
        xl = torch.from_numpy(xg[lo:hi]).to(store.device)
        store[:hi - lo].copy_(xl)                                 # main stream: "step_1" writes p
        call("gkoc_stream_fork_wait", sst, word, C.c_uint32(k))   # side: waits for the product's start
        with torch.cuda.stream(side):
            store[f["halo_base"]:].copy_(halo)                    # the "exchange"
        be.gate_open(side, gate)
        if with_dot:
            be.spmv_gated_dot(local, nl, store, y, gate, dot, fork=(word, C.c_uint32(k)))
        else:
            be.spmv_gated(local, nl, store, y, gate, fork=(word, C.c_uint32(k)))
        torch.cuda.synchronize()
        assert np.array_equal(y.to_numpy()[:, 0], ref), k
        assert int(word[0].item()) == k and int(gate[0][1].item()) == 0
        if with_dot:
            want = float(np.dot(xg[lo:hi], ref))
            assert abs(float(dot.to_numpy()[0, 0]) - want) <= 1e-13 * float(np.dot(np.abs(xg[lo:hi]), np.abs(ref)))


@pytest.mark.gpu
def test_big_slabs_take_the_join_based_product(gexec, oracle):
    """config 3's slab has 512^2-row planes: 2 x 4096 boundary waves, more than may wait on the device
    at once.  The one-kernel product refuses it (gkoc_csr_spmv_gated_fits / GKOC_E_NOT_SUPPORTED) and
    DistributedMatrix falls back to the stream-ordered product; checked on a thin slab of that
    cross-section."""
    import ctypes as C
    import torch
    import ginkgo_amd as g
    import ginkgo_amd.distributed as gd
    from ginkgo_amd import _lib
    L = _lib.lib()
    assert L.gkoc_csr_spmv_gated_fits(C.c_int64(8 * 256 * 256), C.c_int64(256 * 256), C.c_int64(256 * 256)) == 1
    assert L.gkoc_csr_spmv_gated_fits(C.c_int64(64 * 512 * 512), C.c_int64(512 * 512), C.c_int64(512 * 512)) == 0
    assert L.gkoc_csr_spmv_gated_fits(C.c_int64(100), C.c_int64(0), C.c_int64(0)) == 0
    # a slab of 4 planes out of 512 x 512 x 12 (the generator takes any plane range of a cube; use a
    # 512-cube's planes 4..8 so that both neighbours exist)
    grid, z0, nz = 512, 4, 4
    n_plane = grid * grid
    owned = g.stencil_csr(gexec, 3, grid, z0=z0, nz=nz)
    be = gd.HipBackend(gexec)
    lo, hi = z0 * n_plane, (z0 + nz) * n_plane
    local, nl, recv_gidx = be.split(owned, lo, hi, grid ** 3)
    f = nl["full"]
    assert f["head"] == n_plane and f["tail"] == n_plane and not f["gated"]
    gate = be.gate_new()
    store = gexec.zeros((f["halo_base"] + recv_gidx.numel(),), torch.float64)
    y = be.vector(hi - lo)
    rc = L.gkoc_csr_spmv_gated_f64_i32(gexec.stream, C.c_int64(hi - lo), C.c_void_p(local.row_ptrs.data_ptr()),
                                       C.c_void_p(local.col_idxs.data_ptr()), C.c_void_p(local.values.data_ptr()),
                                       C.c_void_p(f["ptrs"].data_ptr()), C.c_void_p(f["cols"].data_ptr()),
                                       C.c_void_p(f["vals"].data_ptr()), C.c_void_p(store.data_ptr()),
                                       C.c_void_p(y.values.data_ptr()), C.c_int64(n_plane), C.c_int64(n_plane),
                                       C.c_void_p(gate[0].data_ptr()), C.c_uint32(1), C.c_void_p(0), C.c_uint32(0))
    assert rc != 0 and b"gkoc_csr_spmv_gated_fits" in L.gkoc_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["torch", "direct"])
def test_collectives_through_rccl_one_rank(how):
    """the same mirror construction with every collective issued through RCCL on the
    device - via torch.distributed backend "nccl", and via the library's own
    communicator (gkoc_comm_*, RcclComm): tests/rccl_mirror_worker.py"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_mirror_worker.py"), "16", how],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + "\n--- stderr ---\n" + p.stderr[-12000:]
    assert "rccl_mirror OK" in p.stdout
