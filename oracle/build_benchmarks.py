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
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
