"""The C++ host-side example (examples/native_cg.cpp) drives CG + block-Jacobi(8)
through the C ABI only.  Compared with the oracle's CG on the same matrix."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "native_cg")


def _run(*args):
    out = subprocess.run([EXE, *map(str, args)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.skipif(not os.path.exists(EXE), reason="examples/native_cg not built (run build())")
def test_native_cg_matches_oracle(oracle):
    grid = 24
    rp, ci, v = oracle.stencil_csr(3, grid)
    n = grid ** 3
    xo, iters, _ = oracle.cg_solve(rp, ci, v, np.ones(n), max_iters=1000, reduction=1e-10,
                                   precond="block", max_block_size=8)
    res = {}
    for mode, lag in (("plain", 0), ("fused", 0), ("fused", 4), ("fused", 7), ("graph", 4)):
        r = _run(grid, 1000, 1e-10, mode, lag)
        assert r["converged"] and abs(r["iterations"] - iters) <= 1
        assert r["true_rel_residual"] <= 1.01e-10
        assert abs(r["x_sum"] - xo.sum()) <= 1e-8 * abs(xo.sum())
        res[(mode, lag)] = r
    # reading the criterion late changes nothing, bit for bit
    assert res[("fused", 0)]["x_sum"] == res[("fused", 4)]["x_sum"] == res[("fused", 7)]["x_sum"]
    assert res[("fused", 0)]["iterations"] == res[("fused", 4)]["iterations"]
    # ... and so does replaying two captured iterations as a hipGraph
    assert res[("graph", 4)]["x_sum"] == res[("fused", 0)]["x_sum"]
    assert res[("graph", 4)]["iterations"] == res[("fused", 0)]["iterations"]
    # iteration limit
    for mode in ("fused", "graph"):
        for cap in (5, 6):
            r = _run(grid, cap, 1e-30, mode, 4)
            assert r["iterations"] == cap and not r["converged"]


DEXE = os.path.join(ROOT, "examples", "native_dist_cg")


def _run_dist(*args, env=None):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PMI_RANK", "PMI_SIZE"):
        e.pop(k, None)
    out = subprocess.run([DEXE, *map(str, args)], capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.skipif(not os.path.exists(DEXE), reason="examples/native_dist_cg not built (run build())")
@pytest.mark.parametrize("solver", ["cg", "pipe_cg"])
def test_native_distributed_driver_matches_oracle(oracle, solver, tmp_path):
    """examples/native_dist_cg.cpp: the row-partitioned CG / PipeCg loop in C++ over gkoc_comm_*.
    One process plays rank 0 of a 2-slab run of the z-mirror-symmetric problem (`mirror`: halo
    exchange and all-reduces go through a real RCCL communicator, the peer's data are its own by
    symmetry) and, as a single rank, the whole problem; both against the single-process oracle."""
    grid = 16
    n, plane = grid ** 3, grid * grid
    rp, ci, v = oracle.stencil_csr(3, grid)
    if solver == "cg":
        xo, iters, _ = oracle.cg_solve(rp, ci, v, np.ones(n), max_iters=1000, reduction=1e-10,
                                       precond="block", max_block_size=8)
    else:
        xo, iters, _ = oracle.krylov_solve("pipe_cg", rp, ci, v, np.ones(n), max_iters=1000,
                                           reduction=1e-10, precond="block")
    got = {}
    for lag in (0, 4):
        dump = str(tmp_path / f"x_{solver}_{lag}")
        r = _run_dist(grid, 1000, 1e-10, solver, lag, "mirror", "dump=" + dump)
        assert r["world"] == 2 and r["mirror"] and r["n_local"] == n // 2
        assert r["converged"] and abs(r["iterations"] - iters) <= 1, (r, iters)
        assert r["true_rel_residual"] <= 2e-10
        x = np.fromfile(dump + ".0", dtype=np.float64)
        assert x.shape == (n // 2,)
        assert np.linalg.norm(x - xo[:n // 2]) <= 1e-8 * np.linalg.norm(xo[:n // 2])
        got[lag] = (r["iterations"], x)
    # reading the criterion late changes nothing, bit for bit
    assert got[0][0] == got[4][0] and np.array_equal(got[0][1], got[4][1])
    # one real rank = the whole domain, no communication
    dump = str(tmp_path / f"x_{solver}_single")
    r = _run_dist(grid, 1000, 1e-10, solver, 4, "dump=" + dump)
    assert r["world"] == 1 and r["converged"] and abs(r["iterations"] - iters) <= 1
    x = np.fromfile(dump + ".0", dtype=np.float64)
    assert np.linalg.norm(x - xo) <= 1e-8 * np.linalg.norm(xo)
    # iteration limit
    r = _run_dist(grid, 7, 1e-30, solver, 4, "mirror")
    assert r["iterations"] == 7 and not r["converged"]


@pytest.mark.skipif(not os.path.exists(DEXE), reason="examples/native_dist_cg not built (run build())")
@pytest.mark.parametrize("solver,world", [("cg", 2), ("pipe_cg", 4), ("cg", 8)])
def test_native_distributed_driver_real_ranks_on_the_mailbox_transport(oracle, solver, world, tmp_path):
    """the same C++ driver as REAL processes - `world` of them sharing cuda:0 - over the library's mailbox
    transport (argument `ipc`: gkoc_comm_ipc_create / _connect, window handles through files): every all-reduce
    and every halo exchange of the loop crosses process boundaries on the device; each rank's slab of x against
    the single-process oracle, the same iteration count on every rank"""
    grid = 16
    n, plane = grid ** 3, grid * grid
    rp, ci, v = oracle.stencil_csr(3, grid)
    if solver == "cg":
        xo, iters, _ = oracle.cg_solve(rp, ci, v, np.ones(n), max_iters=1000, reduction=1e-10,
                                       precond="block", max_block_size=8)
    else:
        xo, iters, _ = oracle.krylov_solve("pipe_cg", rp, ci, v, np.ones(n), max_iters=1000,
                                           reduction=1e-10, precond="block")
    dump = str(tmp_path / f"x_{solver}_{world}")
    idfile = str(tmp_path / "rendezvous")
    procs = []
    for r in range(world):
        e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0",
                 GKOC_ID_FILE=idfile, GKOC_IPC_PATIENCE_MS="60000", GKOC_ARENA_MAX_WALK="24")
        for k in ("PMI_RANK", "PMI_SIZE"):
            e.pop(k, None)
        procs.append(subprocess.Popen([DEXE, str(grid), "1000", "1e-10", solver, "4", "ipc", "dump=" + dump],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e))
    outs = []
    try:
        for p in procs:
            so, se = p.communicate(timeout=300)
            outs.append((p.returncode, so, se))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rc, so, se in outs:
        assert rc == 0, so[-1500:] + se[-3000:]
    its = set()
    planes = [(grid // world) * r + min(r, grid % world) for r in range(world + 1)]
    for r, (rc, so, se) in enumerate(outs):
        line = json.loads([ln for ln in so.strip().splitlines() if ln.startswith("{")][-1])
        assert line["world"] == world and line["rank"] == r and line["converged"]
        its.add(line["iterations"])
        x = np.fromfile(dump + f".{r}", dtype=np.float64)
        lo, hi = planes[r] * plane, planes[r + 1] * plane
        assert x.shape == (hi - lo,)
        assert np.linalg.norm(x - xo[lo:hi]) <= 1e-8 * np.linalg.norm(xo[lo:hi])
    assert len(its) == 1 and abs(its.pop() - iters) <= 1
