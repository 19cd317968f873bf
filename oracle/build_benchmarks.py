#!/usr/bin/env python3
"""TEST / BENCHMARK INFRASTRUCTURE -- Ginkgo's OWN benchmark drivers, unmodified
(benchmark/spmv/spmv.cpp, benchmark/solver/solver.cpp with benchmark/utils/*.hpp), built for
double precision against the drop-in backend (oracle/_ref/dropin/libginkgo_hip.so) the way
benchmark/CMakeLists.txt:90-141 does (-DGKO_BENCHMARK_USE_DOUBLE_PRECISION, no vendor linops).
gflags and nlohmann-json are fetched by the reference's CMake and are not in the image:
tests/dropin/bench_shim/{gflags/gflags.h, nlohmann/json.hpp} (own code) stand in for them.

Outputs: oracle/_ref/dropin/benchmark/{spmv, solver}; run e.g.
  echo '[{"stencil": "27pt", "size": 100}]' | oracle/_ref/dropin/benchmark/spmv -executor hip -formats csr,ell
(tests/test_benchmark_harness_gpu.py, tools/run_gko_benchmarks.sh)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFB = os.path.join(HERE, "_ref")
DROP = os.path.join(REFB, "dropin")
OUT = os.path.join(DROP, "benchmark")


def main():
    ref = os.environ.get("GKO_REFERENCE_DIR", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "benchmark")):
        print("[build_benchmarks] reference not found; keeping prebuilt outputs")
        return 0
    if not os.path.exists(os.path.join(DROP, "include", "ginkgo", "ginkgo.hpp")):
        print("[build_benchmarks] run build_dropin.py first")
        return 1
    os.makedirs(OUT, exist_ok=True)
    cdna_dir = os.path.join(ROOT, "ginkgo_amd", "lib")
    inc = [f"-I{ROOT}/tests/dropin/bench_shim", f"-I{DROP}/include", f"-I{REFB}/include",
           f"-I{ref}/include", f"-I{ref}"]
    flags = ["-std=c++17", "-O2", "-w", "-DGKO_BENCHMARK_USE_DOUBLE_PRECISION"]
    link = [f"-L{DROP}", f"-L{REFB}/lib", "-lginkgo", "-lginkgo_omp", "-lginkgo_reference",
            "-lginkgo_hip", "-lginkgo_cuda", "-lginkgo_dpcpp", "-lginkgo_device", "-fopenmp",
            f"-L{cdna_dir}", "-lgko_cdna4", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,$ORIGIN/../../lib",
            "-Wl,-rpath,$ORIGIN/../../../../ginkgo_amd/lib"]
    progs = {"spmv": "benchmark/spmv/spmv.cpp", "solver": "benchmark/solver/solver.cpp"}
    failed = False
    for name, rel in progs.items():
        p = subprocess.run(["g++"] + flags + inc + [os.path.join(ref, rel), "-o", os.path.join(OUT, name)] + link,
                           capture_output=True, text=True)
        if p.returncode:
            print(f"[build_benchmarks] {name} failed:\n{p.stderr[-8000:]}", file=sys.stderr)
            failed = True
    if not failed:
        print(f"[build_benchmarks] built {', '.join(progs)} in {OUT}")
    failed = build_distributed(ref, flags) or failed
    return 1 if failed else 0


def build_distributed(ref, flags):
    """benchmark/spmv/distributed/spmv.cpp and benchmark/solver/distributed/solver.cpp (unmodified), as
    benchmark/CMakeLists.txt:147-160 builds them when GINKGO_BUILD_MPI is on: against the GPU-aware core of
    oracle/_ref/mpi_ga (build_ref_mpi.py), the drop-in libginkgo_hip.so and, in FRONT of libmpi, the MPI layer
    libgkoc_mpi_rccl.so (the core hands device pointers to MPI; the layer routes them over RCCL or stages
    them).  Outputs: oracle/_ref/mpi_ga/benchmark/{spmv_distributed, solver_distributed}; run with
    mpiexec -n 3 (benchmark/test/spmv_distributed.py, solver_distributed.py: num_procs=3)."""
    mpi_root = os.environ.get("GKO_MPI_ROOT", "/opt/conda")
    ga = os.path.join(REFB, "mpi_ga")
    if not os.path.exists(os.path.join(ga, "lib", "libginkgo.so")) or \
            not os.path.exists(os.path.join(ROOT, "ginkgo_amd", "lib", "libgkoc_mpi_rccl.so")):
        print("[build_benchmarks] no GPU-aware MPI build of the core (build_ref_mpi.py / build_mpi_dropin.py): "
              "distributed drivers skipped")
        return False
    out = os.path.join(ga, "benchmark")
    os.makedirs(out, exist_ok=True)
    inc = [f"-I{ROOT}/tests/dropin/bench_shim", f"-I{ga}/include", f"-I{ref}/include", f"-I{ref}",
           "-idirafter", f"{mpi_root}/include"]
    link = [f"-L{ROOT}/ginkgo_amd/lib", "-Wl,--no-as-needed", "-lgkoc_mpi_rccl", "-Wl,--as-needed",
            f"-L{ga}/lib", "-lginkgo", f"-L{DROP}", "-lginkgo_hip", f"-L{REFB}/lib", "-lginkgo_omp",
            "-lginkgo_reference", "-lginkgo_cuda", "-lginkgo_dpcpp", "-lginkgo_device",
            os.path.join(ga, "lib", "libmpi.so.12"), "-lgko_cdna4", "-fopenmp",
            "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,$ORIGIN/../../dropin", "-Wl,-rpath,$ORIGIN/../../lib",
            "-Wl,-rpath,$ORIGIN/../../../../ginkgo_amd/lib"]
    flags = flags + ["-DHAS_MPI_TIMER=1"]          # benchmark/CMakeLists.txt:121-122
    progs = {"spmv_distributed": "benchmark/spmv/distributed/spmv.cpp",
             "solver_distributed": "benchmark/solver/distributed/solver.cpp"}
    failed = False
    for name, rel in progs.items():
        # (benchmark/utils/mpi_timer.cpp: the timer that reports the longest duration over the ranks - part of
        # the ginkgo_benchmark_cpu_timer library of benchmark/CMakeLists.txt:10-20)
        p = subprocess.run(["g++"] + flags + inc + [os.path.join(ref, rel),
                                                    os.path.join(ref, "benchmark", "utils", "mpi_timer.cpp"),
                                                    "-o", os.path.join(out, name)] + link,
                           capture_output=True, text=True)
        if p.returncode:
            print(f"[build_benchmarks] {name} failed:\n{p.stderr[-8000:]}", file=sys.stderr)
            failed = True
    if not failed:
        print(f"[build_benchmarks] built {', '.join(progs)} in {out}")
    return failed


if __name__ == "__main__":
    sys.exit(main())
