#!/usr/bin/env python3
"""bench.py - the hot-path benchmark (BASELINE.json metric).

Metric : CSR SpMV GB/s on the 27-pt 3-D Laplacian 256^3, fp64 values / int32
         indices (configs[1]); one "step" = one y = A x over the whole matrix
         with all operands resident in HBM.  GB/s = ALGORITHMIC bytes / time:
         nnz*(8+4) + (n+1)*4 + 8*n (x once) + 8*n (y)   (SURVEY.md 8(d)).
Extras : CG iterations/s for configs[2] (CG + block-Jacobi(8), same matrix),
         `roofline` for the SpMV kernel (HIP events on the launch stream),
         `cpu_baseline` (the reference's OmpExecutor from oracle/_ref when
         present, else the plain-C oracle) on a bounded sample.
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


def _interleave_host_memory(on):
    """The reference's arrays are first touched by ONE thread (the copy into the
    executor), i.e. they would all sit on one NUMA node while OpenMP threads of every
    socket read them.  set_mempolicy(MPOL_INTERLEAVE) over all nodes for the
    allocations of the baseline spreads the pages instead.  Returns what was done."""
    import ctypes as C
    try:
        nodes = [d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()]
        n = len(nodes)
        if n < 2:
            return f"{max(n, 1)} NUMA node"
        libc = C.CDLL(None, use_errno=True)
        mask = C.c_ulong((1 << n) - 1 if on else 0)
        mode = 3 if on else 0          # MPOL_INTERLEAVE / MPOL_DEFAULT
        rc = libc.syscall(238, C.c_int(mode), C.byref(mask), C.c_ulong(64))   # SYS_set_mempolicy, x86-64
        if rc != 0:
            return f"{n} NUMA nodes, first touch by one thread (set_mempolicy refused)"
        return f"pages interleaved over {n} NUMA nodes"
    except Exception:
        return "unknown"


def cpu_baseline(grid, csr_host=None, budget_s=12.0):
    """Time the CPU reference on rank 0 on the SAME matrix as the GPU run
    (`csr_host` = (row_ptrs, cols, vals) copied back from the device; generated
    by the oracle when absent): 27-pt `grid`^3 CSR SpMV (fp64/int32).
    kind = "reference": gko::OmpExecutor from the unmodified reference built
    into oracle/_ref (strategy classical); kind = "port": the sequential plain-C
    oracle.  The vectors are allocated once; only `apply` calls are timed."""
    import ctypes as C
    import numpy as np
    from oracle import gko_oracle as o
    numa = _interleave_host_memory(True)
    own = csr_host is not None
    if csr_host is None:
        csr_host = o.stencil_csr(3, grid)
    row_ptrs, cols, vals = csr_host
    n, nnz = len(row_ptrs) - 1, len(vals)
    b = np.random.default_rng(42).uniform(-1, 1, n).reshape(n, 1)
    out = np.zeros((n, 1))
    nbytes = spmv_algorithmic_bytes(n, n, nnz)
    kind, cores, fn, keep = "port", 1, None, None
    try:
        from oracle import ref_shim
        if ref_shim.available():
            cores = os.cpu_count() or 1
            keep = ref_shim.CsrHandle("omp", row_ptrs, cols, vals)
            rl = ref_shim.lib()
            pb, po = b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)
            one = C.c_int64(1)
            fn = lambda: rl.ref_csr_spmv(keep.h, pb, one, po, one, one)
            kind = "reference"
    except Exception:
        fn = None
    if fn is None:
        fn = lambda: o.csr_spmv(row_ptrs, cols, vals, b[:, 0])
    fn()
    t0 = time.perf_counter()
    reps = 0
    while True:
        fn()
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 200:
            break
    _interleave_host_memory(False)
    return {"value": round(nbytes * reps / el / 1e9, 3), "unit": "GB/s",
            "cores": cores, "kind": kind, "cpu": _cpu_model(), "numa": numa,
            "sample": f"27-pt {grid}^3 CSR SpMV fp64/int32"
                      f"{' (the matrix of the GPU run, copied to the host)' if own else ''}, "
                      f"{reps} reps in {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--cg-iters", type=int, default=100,
                    help="fixed CG iterations timed for the iters/s figure")
    ap.add_argument("--cpu-grid", type=int, default=0,
                    help="0 = time the CPU reference on the GPU run's own matrix; "
                         "otherwise a separately generated grid^3 matrix")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--pipe-cg", action="store_true",
                    help="distributed runs: also time PipeCg + block-Jacobi(8) (one all-reduce per "
                         "iteration) and report pipe_cg_iters_per_s next to cg_iters_per_s")
    ap.add_argument("--arena", type=int, default=None,
                    help="GKOC_ARENA mode of the library's allocator: 2 = memory-class regions "
                         "(default), 1 = plain chunks, 0 = one hipMalloc per array (DESIGN.md 3.2)")
    args = ap.parse_args()
    if args.arena is not None:
        os.environ["GKOC_ARENA"] = str(args.arena)

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
                         "with torch.distributed.run --nproc-per-node N")
    # one rank per GPU over RCCL.  GKO_BENCH_BACKEND=gloo (ranks may then share a
    # device, halo/all-reduce staged through the host) exists only to exercise
    # this N > 1 code path on a single-GPU box; its numbers mean nothing.
    backend = os.environ.get("GKO_BENCH_BACKEND", "nccl")
    dev_id = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_id)
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
        op = gd.DistributedStencil(ex, part, rank)
        nnz_global = op.global_nnz
        n_local = op.n_local
        if args.cg_iters > 0:
            t_setup = op.prepare_cg(args.cg_iters, barrier)
        x = op.random_vector(42)
        y = op.zeros_vector()
        step = lambda: op.apply(x, y)

    total_bytes = spmv_algorithmic_bytes(n_global, n_global, nnz_global)

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
        cg = {"cg_iters_per_s": round(iters / t_cg, 2), "cg_iterations": iters,
              "cg_ms_per_iter": round(t_cg * 1e3 / iters, 4),
              "cg_model_gbs": round(cg_bytes * iters / t_cg / 1e9, 1),
              "cg_precond": "block-Jacobi(8)", "cg_setup_s": round(t_setup, 3)}
        if use_dist and args.pipe_cg:
            op.prepare_pipe_cg(args.cg_iters, barrier)
            p_iters, t_p = op.timed_pipe_cg(barrier)
            tp = torch.tensor([t_p], dtype=torch.float64, device=ex.device)
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            cg["pipe_cg_iters_per_s"] = round(p_iters / float(tp.item()), 2)
            cg["pipe_cg_iterations"] = p_iters

    if rank == 0:
        per_gpu_bytes = total_bytes / world
        achieved = per_gpu_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes over this
        # kernel (counters cannot be read from inside the process): the committed
        # summary of the last such run, NOT a measurement of this run
        traffic, traffic_source = None, None
        prof = os.path.join(ROOT, "profiles", "spmv_pmc_latest.json")
        if not use_dist and grid == 256 and os.path.exists(prof):
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
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": ("csr_spmv_pipe3_kernel<double,int,...>" if not use_dist else
                                    "per-rank distributed apply: halo pack + exchange || local "
                                    "csr_spmv_pipe3_kernel, then boundary rows"),
                         "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": int(per_gpu_bytes)},
        }
        out.update(cg)
        if not use_dist:
            import ctypes as C
            from ginkgo_amd import _lib

            class ArenaInfo(C.Structure):
                _fields_ = [("mode", C.c_int32), ("num_classes", C.c_int32),
                            ("chunk_bytes", C.c_int64), ("num_chunks", C.c_int64),
                            ("reserved_bytes", C.c_int64), ("used_bytes", C.c_int64),
                            ("num_allocations", C.c_int64), ("probes", C.c_int64),
                            ("granules_walked", C.c_int64), ("spare_bytes", C.c_int64),
                            ("class_reserved_bytes", C.c_int64 * 3),
                            ("class_used_bytes", C.c_int64 * 3)]
            info = ArenaInfo()
            _lib.call("gkoc_arena_stats", C.byref(info))
            out["placement"] = {
                "note": "device allocator of the library (csrc/arena.hip): one region per memory "
                        "class of the MI355X, class of every 1 GiB granule measured by a probe; "
                        "nothing tuned per run",
                "arena_mode": info.mode, "memory_classes_found": info.num_classes,
                "class_of": {"values": ex.memory_class(a.values), "col_idxs": ex.memory_class(a.col_idxs),
                             "row_ptrs": ex.memory_class(a.row_ptrs), "x": ex.memory_class(x.values),
                             "y": ex.memory_class(y.values)},
                "reserved_gib": round(info.reserved_bytes / 2 ** 30, 2),
                "used_gib": round(info.used_bytes / 2 ** 30, 2),
                "spare_gib": round(info.spare_bytes / 2 ** 30, 2),
                "granules_walked": info.granules_walked, "probe_launches": info.probes}
        if not args.no_cpu and not use_dist:
            if args.cpu_grid:
                out["cpu_baseline"] = cpu_baseline(args.cpu_grid)
            else:
                host = tuple(t.cpu().numpy() for t in (a.row_ptrs, a.col_idxs, a.values))
                out["cpu_baseline"] = cpu_baseline(grid, host)
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
