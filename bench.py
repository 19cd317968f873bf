#!/usr/bin/env python3
"""bench.py - the hot-path benchmark (BASELINE.json metric).

Metric : CSR SpMV GB/s on the 27-pt 3-D Laplacian 256^3, fp64 values / int32
         indices (configs[1]); one "step" = one y = A x over the whole matrix
         with all operands resident in HBM.  GB/s = ALGORITHMIC bytes / time:
         nnz*(8+4) + (n+1)*4 + 8*n (x once) + 8*n (y)   (SURVEY.md 8(d)).
Extras : CG iterations/s for configs[2] (CG + block-Jacobi(8), same matrix),
         `roofline` for the SpMV kernel (HIP events on the launch stream),
         `cpu_baseline` (the reference's OmpExecutor from oracle/_ref when
         present, else the plain-C oracle) on a bounded sample, and `ginkgo_api`:
         the same SpMV and CG through the UNMODIFIED Ginkgo core on the drop-in
         backend (gko::HipExecutor, tests/dropin/dropin_bench.cpp).
N > 1  : the 256^3 problem is row-partitioned into z-slabs (strong scaling),
         one process per GPU, halo exchange + all-reduce over RCCL.

usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--grid G]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def _sig3(v, digits=3):
    """significant digits, not decimals (a per-rank fraction of 4e-5 must not print as 0.0)"""
    return float(f"{v:.{digits}g}")


def spmv_algorithmic_bytes(n_rows, n_cols, nnz, val_bytes=8, idx_bytes=4):
    return nnz * (val_bytes + idx_bytes) + (n_rows + 1) * idx_bytes + \
        n_cols * val_bytes + n_rows * val_bytes


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    """(physical cores, hardware threads) of this host from /proc/cpuinfo"""
    cores, threads, phys, core = set(), 0, None, None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                threads += 1
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
                cores.add((phys, core))
    except OSError:
        pass
    threads = threads or (os.cpu_count() or 1)
    return (len(cores) or threads), threads


def cpu_baseline_child(grid, budget_s):
    """`bench.py --cpu-baseline-child GRID BUDGET` (a process of its own, so that the OpenMP
    run-time reads OMP_NUM_THREADS / OMP_PROC_BIND / OMP_PLACES before anything else started it):
    gko::OmpExecutor's Csr::apply of the unmodified reference (oracle/_ref) on the 27-pt grid^3
    matrix, generated into the executor's arrays by the threads that multiply the rows later
    (parallel first touch, oracle/ref_shim.cpp ref_omp_spmv_bench).  Prints one JSON object."""
    import ctypes as C
    from oracle import ref_shim
    rl = ref_shim.lib()
    rl.ref_omp_spmv_bench.restype = C.c_double
    reps, thr, tri, setup = C.c_int64(0), C.c_int(0), C.c_double(0), C.c_double(0)
    gbs = rl.ref_omp_spmv_bench(C.c_int64(grid), C.c_double(budget_s), C.c_int64(400), C.byref(reps),
                                C.byref(thr), C.byref(tri), C.byref(setup))
    print(json.dumps({"gbs": gbs, "reps": reps.value, "threads": thr.value, "triad_gbs": tri.value,
                      "setup_s": setup.value}))


def cpu_baseline(grid, csr_host=None, budget_s=12.0):
    """The CPU figure beside the GPU one (a reported baseline, not the target): the reference's
    OWN OpenMP backend - gko::OmpExecutor, omp/matrix/csr_kernels.cpp:86-225, strategy classical -
    from the unmodified reference built into oracle/_ref, on the same 27-pt `grid`^3 fp64/int32
    matrix, run the way an OpenMP code should be: one thread per PHYSICAL core, threads bound
    (OMP_PROC_BIND=spread, OMP_PLACES=cores), every array first touched by the thread that reads it.
    The STREAM triad of the same threads is reported next to it (`host_triad_gbs`): it is what this
    host's memory gives an OpenMP loop, `frac_of_host_triad` says how much of that the reference's
    SpMV reaches.  Falls back to the sequential plain-C oracle (kind "port") without oracle/_ref."""
    import subprocess
    cores, hw_threads = _physical_cores()
    try:
        from oracle import ref_shim
        have_ref = ref_shim.available()
    except Exception:
        have_ref = False
    if have_ref:
        env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="spread", OMP_PLACES="cores")
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child",
                                str(grid), str(budget_s)], capture_output=True, text=True,
                               timeout=600, env=env, cwd=ROOT)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
            r = json.loads(line)
            return {"value": round(r["gbs"], 3), "unit": "GB/s", "cores": r["threads"],
                    "kind": "reference", "through": "gko::OmpExecutor of oracle/_ref (the unmodified reference)",
                    "cpu": _cpu_model(),
                    "threads": f"{r['threads']} OpenMP threads = physical cores ({hw_threads} hardware "
                               "threads on the host), OMP_PROC_BIND=spread OMP_PLACES=cores",
                    "numa": "parallel first touch: matrix, b and c are written by the threads that "
                            "read them (static schedule over rows)",
                    "host_triad_gbs": round(r["triad_gbs"], 1),
                    "frac_of_host_triad": round(r["gbs"] / r["triad_gbs"], 3) if r["triad_gbs"] else None,
                    "sample": f"27-pt {grid}^3 CSR SpMV fp64/int32 (same generator as the GPU run, "
                              f"index-exact), {r['reps']} reps in a {budget_s:.0f} s budget, "
                              f"set-up {r['setup_s']:.1f} s"}
        except Exception as e:      # noqa: BLE001 - the baseline must not take the bench line down
            print(f"[bench] OpenMP baseline failed ({e}); falling back to the sequential port",
                  file=sys.stderr)
    import numpy as np
    from oracle import gko_oracle as o
    g = min(grid, 128)              # the sequential port: a bounded sample
    row_ptrs, cols, vals = o.stencil_csr(3, g)
    n, nnz = len(row_ptrs) - 1, len(vals)
    b = np.random.default_rng(42).uniform(-1, 1, n)
    nbytes = spmv_algorithmic_bytes(n, n, nnz)
    o.csr_spmv(row_ptrs, cols, vals, b)
    t0, reps = time.perf_counter(), 0
    while True:
        o.csr_spmv(row_ptrs, cols, vals, b)
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 200:
            break
    return {"value": round(nbytes * reps / el / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
            "cpu": _cpu_model(), "sample": f"27-pt {g}^3 CSR SpMV fp64/int32, sequential plain-C "
                                           f"oracle, {reps} reps in {el:.1f} s"}


def pmc_traffic(grid):
    """HBM bytes per launch of the SpMV kernel from the chip's counters, collected NOW: three
    `rocprofv3 --pmc` passes (kernel-trace only, one counter group each, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes) over a short child run of this script that
    only does the SpMV.  read bytes = RDREQ_128B * 128 + RDREQ_64B * 64 + the remaining requests
    * 32; written bytes = WRREQ_64B * 64 + the remaining * 32 (FETCH_SIZE under-reports on gfx950).
    None if rocprofv3 is not there or a pass fails (the committed summary of the last counter run
    is then used and labelled as such)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3")
    if prof is None or os.environ.get("GKO_BENCH_NO_PMC") == "1" or \
            any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):    # (already under a profiler)
        return None
    groups = (("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"),
              ("TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum"),
              ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"))
    mean = {}
    tmp = tempfile.mkdtemp(prefix="gko_pmc_", dir="/tmp")
    try:
        for i, grp in enumerate(groups):
            out = os.path.join(tmp, f"p{i}")
            cmd = [prof, "--pmc", *grp, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--grid", str(grid), "--steps", "4", "--warmup", "2",
                   "--cg-iters", "0", "--no-cpu", "--no-ginkgo-api", "--no-pmc"]
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd="/tmp",
                               env=dict(os.environ, TMPDIR="/tmp"))
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None
            acc = {}
            for row in csv.DictReader(open(files[0])):
                if "csr_spmv_pipe3_kernel" in row.get("Kernel_Name", ""):
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for c in grp:
                if not acc.get(c):
                    return None
                mean[c] = sum(acc[c]) / len(acc[c])
    except Exception:       # noqa: BLE001 - counters are an extra, never the line's problem
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    r128, r64, rall = mean["TCC_EA0_RDREQ_128B_sum"], mean["TCC_EA0_RDREQ_64B_sum"], mean["TCC_EA0_RDREQ_sum"]
    w64, wall = mean["TCC_EA0_WRREQ_64B_sum"], mean["TCC_EA0_WRREQ_sum"]
    rd = r128 * 128 + r64 * 64 + max(rall - r128 - r64, 0.0) * 32
    wr = w64 * 64 + max(wall - w64, 0.0) * 32
    return {"hbm_bytes_per_launch": int(rd + wr), "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr),
            "counters_mean_per_launch": {k: round(v, 1) for k, v in mean.items()}}


def ginkgo_api_bench(grid, steps, cg_iters):
    """The same two figures through the UNMODIFIED Ginkgo core on the drop-in backend
    (gko::matrix::Csr::apply and gko::solver::Cg + gko::preconditioner::Jacobi(8) on
    gko::HipExecutor, timed with gko::Timer; tests/dropin/dropin_bench.cpp): the north-star product
    is the gko::Executor path, bench.py's own line goes Python -> ctypes -> C ABI.  A process of
    its own; None when the binary has not been built (needs the reference's headers)."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "dropin", "dropin_bench")
    if not os.path.exists(exe):
        return None
    out = {}
    # GKOC_TUNE_DEFERRED_FUSION: 0 = by-products (the default: nothing held back, cg::step_2 leaves
    # ||r|| and the block-Jacobi application <r, z> behind), 2 = one kernel per call, 1 = calls held
    # back and fused (opt-in)
    for fused in (0, 2, 1):
        env = dict(os.environ, GKOC_TUNE_5=str(fused))
        try:
            p = subprocess.run([exe, str(grid), str(steps), str(cg_iters), "--json"], capture_output=True,
                               text=True, timeout=600, env=env, cwd=os.path.dirname(exe))
            r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:      # noqa: BLE001
            out["error"] = f"dropin_bench failed: {e}"
            break
        if fused == 0:
            nbytes = spmv_algorithmic_bytes(r["n"], r["n"], r["nnz"])
            out.update({"through": "gko::HipExecutor (drop-in libginkgo_hip.so), gko::Timer",
                        "csr_apply_ms": r["csr_apply_ms"],
                        "csr_apply_gbs": round(nbytes / r["csr_apply_ms"] / 1e6, 1),
                        "frac": _sig3(nbytes / r["csr_apply_ms"] / 1e6 / HBM_PEAK_GBS),
                        "cg_iters_per_s": r["cg_iters_per_s"], "cg_ms_per_iter": r["cg_ms_per_iter"],
                        "cg_iterations": r["cg_iterations"], "memory_classes": r["memory_classes"],
                        "fused_across_calls": False,
                        "note": "default: no call is held back; ||r|| and <r,z> come with cg::step_2 and "
                                "the block-Jacobi application (gko_binding/fusion.cpp, by-products)"})
        elif fused == 2:
            out["one_kernel_per_call"] = {"cg_iters_per_s": r["cg_iters_per_s"],
                                          "note": "GKOC_TUNE_DEFERRED_FUSION=2"}
        else:
            out["with_fusion_across_calls"] = {"csr_apply_ms": r["csr_apply_ms"],
                                               "cg_iters_per_s": r["cg_iters_per_s"],
                                               "note": "opt-in GKOC_TUNE_DEFERRED_FUSION=1 "
                                                       "(INTEGRATION.md): cg::step_2 + Jacobi + dots "
                                                       "held and run as one kernel"}
    return out


def measured_peaks(g, ex, torch, a, y, reps=10):
    """SURVEY 8(d)'s second denominator, measured in THIS run on THIS device over the SAME buffers the SpMV
    reads: (1) a streaming read - the library's probe kernel (gkoc_arena_probe: every wavefront streams a
    private contiguous 64 KiB piece with 16-byte loads and writes 1 KiB) over the matrix's value array (3.6
    GB at 256^3, far beyond the 256 MB memory-side cache); best of `reps` launches, bytes = read + written;
    (2) the 2-norm of the same array (a reduction kernel of the product path, mean of `reps`); (3) a triad
    y += alpha x on two 2 GiB vectors (24 bytes per element).  The ceiling quoted is the best of (1), (2)."""
    import ctypes as C
    from ginkgo_amd import _lib
    nnz = a.values.numel()
    es = a.values.element_size()
    out = {}
    best = 0.0
    for read_kb in (32, 64):
        waves = nnz * es // (read_kb * 1024)
        x_bytes = waves * read_kb * 1024
        if waves * 1024 > y.values.numel() * y.values.element_size():
            continue
        ns = C.c_int64(0)
        with torch.cuda.device(ex.device):
            _lib.call("gkoc_arena_probe", C.c_void_p(a.values.data_ptr()), C.c_size_t(x_bytes),
                      C.c_void_p(y.values.data_ptr()), C.c_int(read_kb), C.c_int(1024), C.c_int(reps), C.byref(ns))
        if ns.value > 0:
            best = max(best, (x_bytes + waves * 1024) / ns.value)
    out["stream_read_gbs"] = round(best, 1)
    v = g.Dense(ex, a.values.view(-1, 1))
    res = g.Dense.create(ex, (1, 1))

    def timed(fn, nbytes):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return nbytes / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9

    out["norm2_gbs"] = round(timed(lambda: v.compute_norm2(res), nnz * es), 1)
    m = min(nnz, 1 << 28)                       # 2 GiB per vector at most
    x1 = g.Dense.create(ex, (m, 1))
    y1 = g.Dense.create(ex, (m, 1))
    x1.fill(1.0)
    y1.fill(0.0)
    alpha = g.Dense.from_numpy(ex, __import__("numpy").array([[1e-9]]))
    out["triad_gbs"] = round(timed(lambda: y1.add_scaled(alpha, x1), 24 * m), 1)
    out["read_bytes"], out["triad_bytes"] = nnz * es, 24 * m
    out["read_gbs"] = max(out["stream_read_gbs"], out["norm2_gbs"])
    del x1, y1
    return out


def gmres_bench(g, ex, a, rhs, sol, barrier, iters=60, krylov_dim=30):
    """Gmres(30) + block-Jacobi(8), fixed iteration count (two restart cycles), the reference's default
    orthogonalisation (modified Gram-Schmidt, core/solver/gmres.cpp:157-300); bytes per iteration by the
    reference's own model (gmres.cpp:427-446) and by what the launched kernels need."""
    solver = (g.Gmres.build().with_krylov_dim(krylov_dim)
              .with_criteria(g.stop.Iteration.build().with_max_iters(iters),
                             g.stop.ResidualNorm.build().with_reduction_factor(1e-30))
              .with_preconditioner(g.Jacobi.build().with_max_block_size(8))
              .on(ex).generate(a))
    solver.apply(rhs, sol.fill(0.0))            # warm-up (allocates the Krylov basis)
    barrier()
    t0 = time.perf_counter()
    solver.apply(rhs, sol.fill(0.0))
    barrier()
    t = time.perf_counter() - t0
    it = solver.num_iterations
    n, nnz, d = a.size[0], a.get_num_stored_elements(), krylov_dim
    storage = 12 * nnz + 4 * (n + 1) + 64.5 * n                     # matrix + block-Jacobi(8)
    model = (2.5 * d + 10.5 + 14.0 / d) * n * 8 + (1 + 1.0 / d) * storage
    # launched kernels, iteration k of a cycle: Jacobi 2n, SpMV 2n, <v0,w> 2n, k fused steps (w, v_i, v_i+1
    # read, w written: 4n), last update 3n, norm 1n, scaling 2n  =>  (10 + 4k) n values; the restart as in the model
    fused = (10 + 2.0 * (d - 1) + (1 + 14.0 / d)) * n * 8 + (1 + 1.0 / d) * storage
    return {"gmres_iters_per_s": round(it / t, 2), "gmres_ms_per_iter": round(t * 1e3 / it, 4),
            "gmres_iterations": it, "gmres_krylov_dim": d, "gmres_ortho": "mgs (the reference's default)",
            "gmres_precond": "block-Jacobi(8)",
            "gmres_model_bytes_per_iter": int(model),
            "gmres_model_frac": _sig3(model * it / t / 1e9 / HBM_PEAK_GBS, 3),
            "gmres_bytes_fused_per_iter": int(fused),
            "gmres_frac": _sig3(fused * it / t / 1e9 / HBM_PEAK_GBS, 3)}


def self_launch(n):
    """re-run this command line under torch.distributed.run --nproc-per-node n"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def matrix_bench(args):
    """BASELINE configs[4] as a configuration: a MatrixMarket file (--matrix, e.g. SuiteSparse
    Janna/Flan_1565) or, without one, the stand-in of that scale (ginkgo_amd/workloads.py); SpMV in
    CSR and SELL-P (matrix::Sellp, slice size 64), CG + block-Jacobi(--block-size) on --format; at
    N ranks the rows are split into contiguous ranges with equal shares of the ENTRIES
    (Partition::build_from_contiguous), halo exchange + all-reduce as in the stencil run.
    Reference: benchmark/utils/generator.hpp, core/base/mtx_io.cpp,
    include/ginkgo/core/distributed/partition.hpp:262."""
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist
    import ginkgo_amd as g
    from ginkgo_amd import distributed as gd
    from ginkgo_amd import workloads as wl

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    backend = os.environ.get("GKO_BENCH_BACKEND", "nccl")
    dev_id = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_id)
    if os.environ.get("GKO_TEST_ONE_CLASS_RANK") == str(rank):
        os.environ["GKOC_ARENA_MAX_CLASSES"] = "1"       # (tests: what one rank's allocator finds is its own)
    ex = g.Cdna4Executor.create(dev_id)
    use_dist = world > 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_id))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    bs = max(1, args.block_size)
    t_load = time.perf_counter()
    if args.matrix:
        # every rank reads the file (a parallel file system serves N readers; the parse is the cost)
        n, n_cols, r_, c_, v_ = wl.read_mtx(args.matrix)
        if n != n_cols:
            raise SystemExit("--matrix: the solver configurations need a square matrix")
        rp, ci, vv = wl.csr_from_triplets(n, n, r_, c_, v_)
        del r_, c_, v_
        prefix = rp.astype(np.int64)
        offsets = wl.partition_by_nnz(prefix, world, align=bs)
        lo, hi = offsets[rank], offsets[rank + 1]
        own = ((rp[lo:hi + 1] - rp[lo]).astype(np.int32), ci[rp[lo]:rp[hi]], vv[rp[lo]:rp[hi]])
        nnz = int(rp[-1])
        name = os.path.basename(args.matrix)
        data = f"file {name}"
        workload = f"{name}: n = {n}, nnz = {nnz}"
        del rp, ci, vv
    elif args.workload == "irregular":
        n = args.irr_n
        prefix = wl.irregular_row_prefix(n)
        nnz = int(prefix[-1])
        offsets = wl.partition_by_nnz(prefix, world, align=bs)
        lo, hi = offsets[rank], offsets[rank + 1]
        own = wl.irregular_rows(n, lo, hi)
        lens = np.diff(prefix)
        data = "synthetic heavy-tailed stand-in for an irregular SuiteSparse matrix"
        workload = (f"irregular stand-in for configs[4] (ginkgo_amd/workloads.py irregular_rows): symmetric positive "
                    f"definite, power-law row lengths + {wl.IRR_HUBS} hub rows, n = {n}, nnz = {nnz}; entries per row: "
                    f"mean {lens.mean():.1f}, median {int(np.median(lens))}, 99.9 % {int(np.percentile(lens, 99.9))}, "
                    f"max {int(lens.max())}, {int((lens > 4096).sum())} rows beyond GKOC_CSR_LONG_ROW = 4096")
        del lens
    else:
        grid = args.flan_grid
        n, nnz = wl.flan_like_dims(grid)
        prefix = wl.flan_like_row_prefix(grid)
        offsets = wl.partition_by_nnz(prefix, world, align=bs if 3 % bs == 0 or bs % 3 == 0 else 3 * bs)
        lo, hi = offsets[rank], offsets[rank + 1]
        own = wl.flan_like_rows(grid, lo, hi)
        data = "synthetic stand-in for Flan_1565"
        workload = (f"stand-in for SuiteSparse Janna/Flan_1565 (file not available offline): "
                    f"L27({grid}^3) (x) B3, n = {n}, nnz = {nnz}, up to 81 per row")
    t_load = time.perf_counter() - t_load
    n_local = hi - lo
    owned = g.Csr.from_arrays(ex, (n_local, n), *own)
    del own
    rng = np.random.default_rng(42)
    xg = rng.uniform(-1, 1, n)

    def time_op(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        barrier()
        wall = time.perf_counter() - t0
        t = torch.tensor([wall], dtype=torch.float64, device=ex.device)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / steps, e0.elapsed_time(e1) / steps

    formats, cg, comm_check = {}, {}, None
    csr_bytes = spmv_algorithmic_bytes(n, n, nnz)
    if not use_dist:
        a = owned
        x = g.Dense.from_numpy(ex, xg)
        y = g.Dense.create(ex, (n, 1))
        ops = {"csr": a, "sellp": a.convert_to_sellp()}
        stored = {"csr": nnz, "sellp": int(ops["sellp"].values.numel())}
        digest = {}
        for name_, op in ops.items():
            wall, kms = time_op(lambda: op.apply(x, y), args.steps, args.warmup)
            nb = csr_bytes if name_ == "csr" else 12 * stored[name_] + 8 * (-(-n // 64) + 1) + 16 * n
            formats[name_] = {"ms": round(kms, 5), "gbs_algorithmic_csr_bytes": round(csr_bytes / kms / 1e6, 1),
                              "frac_of_own_bytes": round(nb / kms / 1e6 / HBM_PEAK_GBS, 4),
                              "stored_over_nnz": round(stored[name_] / nnz, 4), "wall_ms": round(wall * 1e3, 5)}
            digest[name_] = y.values.clone()
        # the two formats add a row's entries in the same order: the same bits
        formats["sellp"]["bit_identical_to_csr"] = bool(torch.equal(digest["csr"], digest["sellp"]))
        if args.cg_iters > 0:
            prec = g.Jacobi.build().with_max_block_size(bs).on(ex).generate(a) if bs > 1 else None
            for name_, op in ops.items():
                b = (g.Cg.build().with_criteria(g.stop.Iteration.build().with_max_iters(args.cg_iters),
                                                g.stop.ResidualNorm.build().with_reduction_factor(1e-30)))
                if prec is not None:
                    b = b.with_generated_preconditioner(prec)
                else:
                    b = b.with_preconditioner(g.Jacobi.build().with_max_block_size(1))
                s_ = b.on(ex).generate(op)
                rhs, sol = g.Dense.from_numpy(ex, np.ones(n)), g.Dense.from_numpy(ex, np.zeros(n))
                s_.apply(rhs, sol)
                barrier()
                sol.fill(0.0)
                t0 = time.perf_counter()
                s_.apply(rhs, sol)
                barrier()
                t_cg = time.perf_counter() - t0
                cg[name_] = {"cg_iters_per_s": round(s_.num_iterations / t_cg, 2), "cg_iterations": s_.num_iterations}
        kernel_ms = formats[args.format]["ms"]
        wall_ms = formats[args.format]["wall_ms"]
    else:
        comm = gd.default_comm(ex)
        be = gd.HipBackend(ex)
        part = gd.Partition(offsets)
        comm_check = gd.comm_self_check(ex, comm, n_elems=4096)
        comm_check["transport_choice"] = dict(gd.default_comm.last)
        for name_ in ("csr", "sellp"):
            dm = gd.DistributedMatrix(be, comm, part, owned, local_format=name_)
            dm.agreed_self_check()             # one-kernel product vs join-based one, all ranks agree
            x = be.vector_from(xg[lo:hi])
            y = be.vector(n_local)
            wall, kms = time_op(lambda: dm.apply(x, y), args.steps, args.warmup)
            formats[name_] = {"ms": round(kms, 5), "wall_ms": round(wall * 1e3, 5),
                              "gbs_algorithmic_csr_bytes": round(csr_bytes / wall / 1e9, 1),
                              "halo_values_in": dm.n_halo, "peers": sum(1 for c in dm.recv_counts if c > 0),
                              "one_kernel_product": dm._gate is not None}
            if args.cg_iters > 0:
                s_ = gd.DistributedCg(be, comm, dm, args.cg_iters, 1e-300, bs if bs > 1 else 1)
                rhs, sol = be.vector_from(np.ones(n_local)), be.vector(n_local)
                s_.apply(rhs, sol)
                barrier()
                sol.fill(0.0)
                t0 = time.perf_counter()
                s_.apply(rhs, sol)
                barrier()
                tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=ex.device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                cg[name_] = {"cg_iters_per_s": round(s_.num_iterations / float(tt.item()), 2),
                             "cg_iterations": s_.num_iterations}
        kernel_ms = formats[args.format]["ms"]
        wall_ms = formats[args.format]["wall_ms"]
    # every rank's share (a slow rank explains itself)
    per_rank = None
    if use_dist:
        mine = torch.tensor([kernel_ms, float(n_local), float(prefix[hi] - prefix[lo])], dtype=torch.float64,
                            device=ex.device)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = [{"rank": r, "kernel_ms": round(float(t[0]), 5), "rows": int(t[1]), "nnz": int(t[2])}
                    for r, t in enumerate(gathered)]
    if rank == 0:
        achieved = csr_bytes / world / (kernel_ms * 1e-3) / 1e9
        info = ex.arena_info()
        out = {"metric": "spmv_effective_bandwidth", "value": round(csr_bytes / (wall_ms * 1e-3) / 1e9, 1),
               "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(wall_ms, 5), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f64", "data": data,
               "config": {"workload": workload, "format": args.format, "value_type": "f64", "index_type": "int32",
                          "partition": "contiguous rows, equal shares of the stored entries" if use_dist else "one domain",
                          "preconditioner": f"block-Jacobi({bs})", "load_s": round(t_load, 2),
                          "memory_classes_found": info["num_classes"],
                          "bytes": "CSR model of SURVEY 8(d): 12 nnz + 4 (n + 1) + 16 n, for every format"},
               "formats": formats, "cg": cg,
               "roofline": {"bound": "hbm", "achieved": _sig3(achieved, 5), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": _sig3(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                            "kernel_ms": round(kernel_ms, 5),
                            "traffic_note": "counters are collected for the 256^3 headline only"},
               "cpu_baseline": {"value": None, "unit": "GB/s", "cores": None, "kind": None, "sample": None,
                                "note": "the CPU twin of this workload is tools/flan_bench.py's scipy product; "
                                        "the bounded OmpExecutor baseline belongs to the headline line"}}
        if per_rank:
            out["roofline"]["per_rank"] = per_rank
            out["comm_check"] = comm_check
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--cg-iters", type=int, default=100,
                    help="fixed CG iterations timed for the iters/s figure")
    ap.add_argument("--gmres-iters", type=int, default=60,
                    help="fixed Gmres(30) + block-Jacobi(8) iterations timed at N = 1 (0 = skip)")
    ap.add_argument("--cpu-grid", type=int, default=0,
                    help="0 = the CPU baseline runs the GPU run's grid; otherwise this one")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not collect the SpMV kernel's HBM traffic with rocprofv3 --pmc passes "
                         "(roofline.traffic then comes from the committed summary of the last counter run)")
    ap.add_argument("--no-ginkgo-api", action="store_true",
                    help="skip the second measurement through the unmodified Ginkgo core")
    ap.add_argument("--cpu-baseline-child", nargs=2, metavar=("GRID", "BUDGET_S"), default=None,
                    help=argparse.SUPPRESS)
    ap.add_argument("--pipe-cg", action="store_true", help="(default for N > 1; kept for old command lines)")
    ap.add_argument("--no-pipe-cg", action="store_true",
                    help="distributed runs: do NOT time PipeCg + block-Jacobi(8) (one all-reduce per "
                         "iteration, overlapped) next to Cg; by default pipe_cg_iters_per_s is reported "
                         "beside cg_iters_per_s")
    ap.add_argument("--matrix", default=None,
                    help="a MatrixMarket file: run configs[4] (SELL-P vs CSR, CG + block-Jacobi) on it")
    ap.add_argument("--workload", default=None, choices=[None, "flan", "irregular"],
                    help="configs[4] without the file: flan = the regular stand-in L27(g^3) (x) B3 (24 - 81 entries per "
                         "row); irregular = the heavy-tailed one (power-law row lengths, hub rows beyond 4096 entries)")
    ap.add_argument("--irr-n", type=int, default=4000000, help="--workload irregular: order of the matrix")
    ap.add_argument("--format", default="csr", choices=["csr", "sellp"],
                    help="--matrix / --workload flan: the format `value` is quoted on")
    ap.add_argument("--flan-grid", type=int, default=80, help="stand-in size: L27(g^3) (x) B3")
    ap.add_argument("--block-size", type=int, default=3, help="--matrix / flan: block-Jacobi block size")
    ap.add_argument("--arena", type=int, default=None,
                    help="GKOC_ARENA mode of the library's allocator: 2 = memory-class regions "
                         "(default), 1 = plain chunks, 0 = one hipMalloc per array (DESIGN.md 3.2)")
    args = ap.parse_args()
    if args.cpu_baseline_child:
        return cpu_baseline_child(int(args.cpu_baseline_child[0]), float(args.cpu_baseline_child[1]))
    if args.arena is not None:
        os.environ["GKOC_ARENA"] = str(args.arena)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, the way the
        # driver does (one process per GPU, rendezvous on 127.0.0.1); rank 0 prints the line
        return self_launch(args.gpus)
    if args.matrix or args.workload in ("flan", "irregular"):
        return matrix_bench(args)

    # stdout carries exactly one line, the JSON result: everything else written to
    # file descriptor 1 by this process (RCCL prints a version banner there from C
    # stdio, flushed at exit, i.e. AFTER the result) is sent to stderr instead
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import ginkgo_amd as g

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}; launch "
                         "with torch.distributed.run --nproc-per-node N (or without a launcher: "
                         "bench.py starts its own ranks)")
    # one rank per GPU over RCCL.  GKO_BENCH_BACKEND=gloo (ranks may then share a
    # device, halo/all-reduce staged through the host) exists only to exercise
    # this N > 1 code path on a single-GPU box; its numbers mean nothing.
    backend = os.environ.get("GKO_BENCH_BACKEND", "nccl")
    dev_id = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_id)
    if os.environ.get("GKO_TEST_ONE_CLASS_RANK") == str(rank):
        os.environ["GKOC_ARENA_MAX_CLASSES"] = "1"       # (tests: what one rank's allocator finds is its own)
    ex = g.Cdna4Executor.create(dev_id)
    # GKO_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, barriers,
    # max-over-ranks, DistributedStencil) with a single rank - checks the RCCL calls
    # of this script on a 1-GPU box
    use_dist = world > 1 or os.environ.get("GKO_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", dev_id))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    grid = args.grid
    n_global = grid ** 3

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Placement is the allocator's job (csrc/arena.hip): matrix values,
    # index arrays and vectors come from three different memory classes of the device;
    # nothing is tuned or re-homed here.
    solver = None
    t_setup = 0.0
    if not use_dist:
        a = g.stencil_csr(ex, 3, grid)
        n_local = n_global
        nnz_global = a.get_num_stored_elements()
        x = g.Dense.from_numpy(
            ex, __import__("numpy").random.default_rng(42).uniform(-1, 1, n_global))
        if args.cg_iters > 0:
            t_setup = time.perf_counter()
            solver = (g.Cg.build()
                      .with_criteria(g.stop.Iteration.build().with_max_iters(args.cg_iters),
                                     g.stop.ResidualNorm.build().with_reduction_factor(1e-30))
                      .with_preconditioner(g.Jacobi.build().with_max_block_size(8))
                      .on(ex).generate(a))
            barrier()
            t_setup = time.perf_counter() - t_setup
            import numpy as np
            rhs = g.Dense.from_numpy(ex, np.ones(n_local))
            sol = g.Dense.from_numpy(ex, np.zeros(n_local))
            solver.apply(rhs, sol.fill(0.0))       # warm-up solve (allocates the workspace)
            barrier()
        y = g.Dense.create(ex, (n_local, 1))
        step = lambda: a.apply(x, y)
        op = a
    else:
        from ginkgo_amd import distributed as gd
        part = gd.SlabPartition(grid, world)

        # a hang cannot be cancelled, but it can be NAMED: the watchdogs of ginkgo_amd.distributed end the
        # process (exit code 86); rank 0 leaves a line that says what was running
        def dying_line(what):
            if rank == 0:
                result_out.write(json.dumps({
                    "metric": f"CSR SpMV GB/s (27-pt 3D Laplacian {grid}^3, fp64/int32)", "value": None,
                    "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "error": f"watchdog: {what} did not finish (a collective hangs); nothing was measured",
                    "comm_check": {"transport_choice": dict(gd.default_comm.last)}}) + "\n")
                result_out.flush()
        gd._Watchdog.on_fire = staticmethod(dying_line)

        class TransportDead(Exception):
            pass

        def agree_min(ok):
            f = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=ex.device)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            return float(f.item()) >= 1.0

        def bring_up(comm):
            """operator, communicator check, product check, warm-up solve on `comm` (None: default_comm
            chooses).  Every failure is agreed on by all ranks before anybody acts on it; a transport whose
            waits ran out of patience is reported as TransportDead on ALL ranks."""
            op = gd.DistributedStencil(ex, part, rank, comm=comm)
            # known-answer test of every collective form the solvers use + their latencies here;
            # wrong data raises on all ranks, a hang ends the job with a message (not a timeout)
            cc = gd.comm_self_check(ex, op.comm, n_elems=grid * grid)
            # which transport carries the data path and what each candidate cost on one Cg iteration's
            # communication (IpcComm: the library's mailboxes in peer-mapped memory; RcclComm: RCCL)
            cc["transport_choice"] = dict(gd.default_comm.last) if comm is None else \
                {"chosen": type(comm).__name__, "why": "second attempt"}
            # who is where: what the communicator itself counted (ncclCommCount), RCCL's version, every
            # rank's device - "did it see N ranks" must be answerable from the line
            cc["topology"] = op.comm.topology() if hasattr(op.comm, "topology") else \
                {"transport": f"torch.distributed ({backend})", "ranks": world, "ranks_seen": dist.get_world_size()}
            # the one-kernel product (boundary waves that wait for their halo inside the kernel) against
            # the join-based one on THIS communicator before anything is timed; a rank that sees a
            # difference, a wave that gave up or a fork that timed out sends ALL ranks to the join-based
            # product (the reference's shape) - the line says which one was measured
            pc = op.matrix.agreed_self_check()
            ts = 0.0
            if args.cg_iters > 0:
                # (the warm-up solve ends with the solvers' own checks - a boundary wave that gave up, a
                # fork that timed out, a mailbox wait that ran out of patience: raised at the END of the
                # solve, after every collective has been issued, so all ranks arrive here and agree)
                err = None
                try:
                    ts = op.prepare_cg(args.cg_iters, barrier)
                except gd.GkoError as e:
                    err = str(e)[:200]
                if not agree_min(err is None):
                    dead = hasattr(op.comm, "status") and op.comm.status() != 0
                    if not agree_min(not dead):
                        raise TransportDead(err or "a wait of the transport ran out of patience on another rank")
                    op.matrix.conservative()
                    pc = dict(pc, one_kernel_product=False,
                              self_check="warm-up solve: " + (err or "failed on another rank"))
                    ts = op.prepare_cg(args.cg_iters, barrier)
            return op, cc, pc, ts

        try:
            op, comm_check, product_check, t_setup = bring_up(None)
        except (gd.GkoError, TransportDead) as e:
            # (both are raised on every rank together.)  The process group that brought the ranks here
            # still works: the data path goes through it - RCCL under torch.distributed on one rank per
            # GPU - and the line says so
            first = f"{type(e).__name__}: {e}"[:300]
            print(f"[bench] rank {rank}: {first} - second attempt on torch.distributed", file=sys.stderr)
            op, comm_check, product_check, t_setup = bring_up(gd.TorchComm())
            comm_check["transport_choice"] = {"chosen": "TorchComm", "first_choice": dict(gd.default_comm.last),
                                              "why": "the first choice failed after it had come up: " + first}
        nnz_global = op.global_nnz
        n_local = op.n_local
        x = op.random_vector(42)
        y = op.zeros_vector()
        step = lambda: op.apply(x, y)

    total_bytes = spmv_algorithmic_bytes(n_global, n_global, nnz_global)
    dist_profile = op.profile(x, y) if use_dist else None

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps
    t = torch.tensor([wall], dtype=torch.float64, device=ex.device)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    ms_per_step = wall * 1e3 / args.steps
    gbs = total_bytes / (wall / args.steps) / 1e9

    # ---- CG + block-Jacobi(8) iterations/s (configs[2]), fixed iteration count
    cg = {}
    if args.cg_iters > 0:
        import numpy as np
        if not use_dist:
            t1 = time.perf_counter()
            solver.apply(rhs, sol.fill(0.0))
            barrier()
            t_cg = time.perf_counter() - t1
            iters = solver.num_iterations
        else:
            iters, t_cg = op.timed_cg(barrier)
        tt = torch.tensor([t_cg], dtype=torch.float64, device=ex.device)
        if use_dist:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_cg = float(tt.item())
        n, nnz = n_global, nnz_global
        cg_bytes = 12 * nnz + 4 * (n + 1) + 64.5 * n + 144 * n  # cg.cpp:133-141 model
        # what the three kernels of an iteration really need (the model's 18 n vector
        # values are those of the UNFUSED loop): SpMV + <p,q> (the product's bytes + p once more),
        # step_2 + block-Jacobi + both sums (64n + 4(n/8+1) blocks, x r p q in, x r z out = 56n), step_1 24n
        cg_fused = (12 * nnz + 4 * (n + 1) + 16 * n + 8 * n) + (64 * n + 4 * (n // 8 + 1) + 56 * n) + 24 * n
        cg = {"cg_iters_per_s": round(iters / t_cg, 2), "cg_iterations": iters,
              "cg_ms_per_iter": round(t_cg * 1e3 / iters, 4),
              "cg_model_gbs": round(cg_bytes * iters / t_cg / 1e9, 1),
              "cg_model_note": "Ginkgo's model of the UNFUSED loop (cg.cpp:133-141, 18 n vector values): an "
                               "upper bound on the bytes moved, not the fraction",
              "cg_bytes_fused": int(cg_fused),
              "cg_fused_gbs": round(cg_fused * iters / t_cg / 1e9 / world, 1),
              "cg_frac": _sig3(cg_fused * iters / t_cg / 1e9 / (HBM_PEAK_GBS * world), 3),
              "cg_precond": "block-Jacobi(8)", "cg_setup_s": round(t_setup, 3)}
        if not use_dist and args.gmres_iters > 0:
            try:
                cg.update(gmres_bench(g, ex, a, rhs, sol, barrier, iters=args.gmres_iters))
            except Exception as e:      # noqa: BLE001 - an extra must not take the line down
                cg["gmres_error"] = f"{type(e).__name__}: {e}"[:300]
        if use_dist and not args.no_pipe_cg:
            # the pipelined CG (core/solver/pipe_cg.cpp:95-297): ONE all-reduce per iteration, travelling
            # while the preconditioner and the SpMV run - the solver for a latency-bound strong-scaling
            # run.  A failure here must not take the line down: it is recorded instead.
            ok = 1
            try:
                op.prepare_pipe_cg(args.cg_iters, barrier)
                p_iters, t_p = op.timed_pipe_cg(barrier)
            except Exception as e:      # noqa: BLE001
                ok, p_iters, t_p = 0, 0, 0.0
                cg["pipe_cg_error"] = f"rank {rank}: {type(e).__name__}: {e}"[:300]
            tp = torch.tensor([t_p, float(ok)], dtype=torch.float64, device=ex.device)
            dist.all_reduce(tp[:1], op=dist.ReduceOp.MAX)
            dist.all_reduce(tp[1:], op=dist.ReduceOp.MIN)
            if float(tp[1].item()) > 0 and p_iters > 0:
                cg["pipe_cg_iters_per_s"] = round(p_iters / float(tp[0].item()), 2)
                cg["pipe_cg_iterations"] = p_iters
                cg["pipe_cg_ms_per_iter"] = round(float(tp[0].item()) * 1e3 / p_iters, 4)
            else:
                cg.setdefault("pipe_cg_error", "failed on another rank")

    # every rank's kernel time and what its allocator found (N > 1: a slow rank explains itself)
    per_rank = None
    if use_dist:
        ai = ex.arena_info()
        mine = torch.tensor([kernel_ms, float(ai["num_classes"]), ai["search_ns"] / 1e6,
                             float(ai["granules_walked"])], dtype=torch.float64, device=ex.device)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = [[round(float(v), 4) for v in t.tolist()] for t in gathered]

    if rank == 0:
        per_gpu_bytes = total_bytes / world
        achieved = per_gpu_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes over this
        # kernel (counters cannot be read from inside the process): the committed
        # summary of the last such run, NOT a measurement of this run
        traffic, traffic_source, traffic_detail = None, None, None
        prof = os.path.join(ROOT, "profiles", "spmv_pmc_latest.json")
        if not use_dist and not args.no_pmc:
            torch.cuda.synchronize()
            traffic_detail = pmc_traffic(grid)
            if traffic_detail is not None:
                traffic = traffic_detail["hbm_bytes_per_launch"]
                traffic_source = ("this run: three rocprofv3 --pmc passes (TCC_EA0_RDREQ / WRREQ by request "
                                  "size) over a child run of the same SpMV on this device, mean per launch")
        if traffic is None and not use_dist and grid == 256 and os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get("hbm_bytes_per_launch")
                traffic_source = "profiles/spmv_pmc_latest.json (separate rocprofv3 --pmc run)"
            except Exception:
                traffic = None
        out = {
            "metric": f"CSR SpMV GB/s (27-pt 3D Laplacian {grid}^3, fp64/int32)",
            "value": round(gbs, 1), "unit": "GB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"27-pt 3D Laplacian {grid}^3 CSR SpMV fp64 "
                                   f"(BASELINE configs[1]); n={n_global}, nnz={nnz_global}",
                       "index_type": "int32", "partition": f"{world} z-slab(s)",
                       **({"communicator": type(op.comm).__name__} if use_dist else {}),
                       "pct_hbm_peak": round(100 * gbs / (HBM_PEAK_GBS * world), 1)},
            "roofline": {"bound": "hbm", "achieved": _sig3(achieved, 5),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": _sig3(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         **({"traffic_over_algorithmic": round(traffic / per_gpu_bytes, 4)} if traffic else {}),
                         **({"traffic_counters": traffic_detail["counters_mean_per_launch"]}
                            if traffic_detail else {}),
                         "kernel": ("csr_spmv_pipe3_kernel<double,int,...>" if not use_dist else
                                    "per-rank distributed apply: halo pack + exchange || local "
                                    "csr_spmv_pipe3_kernel, then boundary rows"),
                         "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": int(per_gpu_bytes)},
        }
        out.update(cg)
        if not use_dist:
            try:
                pk = measured_peaks(g, ex, torch, a, y)
                out["roofline"]["peak_measured"] = pk["read_gbs"]
                out["roofline"]["frac_of_measured"] = _sig3(achieved / pk["read_gbs"], 4)
                out["roofline"]["peak_measured_how"] = (
                    f"same run, same device, the SpMV's own buffers: streaming read of the matrix's value array "
                    f"({pk['read_bytes']} bytes; the library's probe kernel, best of 10 launches) "
                    f"{pk['stream_read_gbs']} GB/s; 2-norm of the same array {pk['norm2_gbs']} GB/s; triad "
                    f"y += a x on two {pk['triad_bytes'] // 24 * 8} byte vectors {pk['triad_gbs']} GB/s")
                out["roofline"]["triad_measured"] = pk["triad_gbs"]
            except Exception as e:      # noqa: BLE001
                out["roofline"]["peak_measured"] = None
                out["roofline"]["peak_measured_error"] = f"{type(e).__name__}: {e}"[:200]
        if use_dist:
            out["comm_check"] = comm_check
            out["distributed_product"] = product_check
            out["rank0_profile"] = dist_profile
            out["roofline"]["per_rank"] = [
                {"rank": r, "kernel_ms": v[0],
                 "achieved": _sig3(per_gpu_bytes / (v[0] * 1e-3) / 1e9) if v[0] > 0 else None,
                 "frac": _sig3(per_gpu_bytes / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if v[0] > 0 else None,
                 "memory_classes_found": int(v[1]), "search_ms": v[2], "granules_walked": int(v[3])}
                for r, v in enumerate(per_rank)]
            out["roofline"]["search_ms_max_over_ranks"] = max(v[2] for v in per_rank)
            out["roofline"]["memory_classes_min_over_ranks"] = int(min(v[1] for v in per_rank))
            out["roofline"]["traffic_note"] = ("counters are collected at N = 1 only (a rocprofv3 --pmc pass "
                                               "per counter group over a child run)")
        info = ex.arena_info()
        placement = {
            "note": "device allocator of the library (csrc/arena.hip): one region per memory "
                    "class of the MI355X, found by ONE galloping walk over 1 GiB granules whose "
                    "class is measured by a probe; nothing tuned per run",
            "arena_mode": info["mode"], "memory_classes_found": info["num_classes"],
            "reserved_gib": round(info["reserved_bytes"] / 2 ** 30, 2),
            "used_gib": round(info["used_bytes"] / 2 ** 30, 2),
            "spare_gib": round(info["spare_bytes"] / 2 ** 30, 2),
            "granules_walked": info["granules_walked"],
            "granules_classified": info["granules_classified"],
            "probe_launches": info["probes"], "probe_retries": info["probe_retries"],
            "search_ms": round(info["search_ns"] / 1e6, 1),
            "search_budget_ms": info["search_budget_ms"], "search_budget_spent": bool(info["search_budget_spent"]),
            "granules_unclassified": info["granules_unclassified"]}
        # what a process pays before its first product: the allocator's searches + the solver's set-up
        out["startup_s"] = round(info["search_ns"] / 1e9 + (t_setup or 0.0), 3)
        if not use_dist:
            placement["class_of"] = {"values": ex.memory_class(a.values), "col_idxs": ex.memory_class(a.col_idxs),
                                     "row_ptrs": ex.memory_class(a.row_ptrs), "x": ex.memory_class(x.values),
                                     "y": ex.memory_class(y.values)}
        else:
            placement["class_of"] = getattr(op.backend, "placement_log", None)
        out["placement"] = placement
        # a slow line must explain itself: the classes belong to the configuration
        out["config"]["memory_classes_found"] = info["num_classes"]
        out["config"]["class_of"] = placement["class_of"]
        if not args.no_ginkgo_api and not use_dist:
            torch.cuda.synchronize()
            api = ginkgo_api_bench(grid, args.steps, max(args.cg_iters, 200) if args.cg_iters > 0 else 20)
            if api is not None:
                out["ginkgo_api"] = api
        cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"gko_bench_cpu_baseline_{grid}.json")
        if not args.no_cpu and not use_dist:
            out["cpu_baseline"] = cpu_baseline(args.cpu_grid or grid)
            try:
                json.dump(out["cpu_baseline"], open(cache, "w"))
            except OSError:
                pass
        elif use_dist:
            # no CPU twin at N > 1 (the baseline is one host running the whole problem): the N = 1
            # figure is carried - this host's, if an N = 1 run left it here, else the committed one
            base, src = None, None
            for path, what in ((cache, "the N = 1 run of this script on this host"),
                               (os.path.join(ROOT, "profiles", "bench_line_latest.json"),
                                "the committed N = 1 line profiles/bench_line_latest.json (another box)")):
                try:
                    d = json.load(open(path))
                    base, src = d.get("cpu_baseline", d), what
                    break
                except (OSError, ValueError):
                    continue
            if base is not None and "value" in base:
                out["cpu_baseline"] = dict(base, note=f"no CPU twin at N > 1; carried from {src}")
            else:
                out["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": None, "kind": None,
                                       "sample": None, "note": "no CPU twin at N > 1 and no N = 1 figure at hand"}
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
