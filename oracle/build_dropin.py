#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- the acceptance programs of the drop-in library, in oracle/_ref/dropin/:

  libginkgo_hip.so   a COPY of the product's ginkgo_amd/lib/libginkgo_hip.so (built by
                     ginkgo_amd/gko_binding/build.py, which this script calls first)
  dropin_test        tests/dropin/dropin_test.cpp linked against the unmodified
                     core (oracle/_ref/lib/libginkgo.so)
  preconditioned-solver   Ginkgo's own example, unmodified, + its data files

Needs /root/reference (headers) and a finished oracle/build_ref.py; on the GPU
box the prebuilt outputs are used as they are."""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFB = os.path.join(HERE, "_ref")
OUT = os.path.join(REFB, "dropin")


def run(cmd):
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        print(" ".join(cmd), file=sys.stderr)
        print(p.stderr[-8000:], file=sys.stderr)
        sys.exit(1)


def main():
    ref = os.environ.get("GKO_REFERENCE_DIR", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "include", "ginkgo")):
        print("[build_dropin] reference headers not found; keeping prebuilt outputs")
        return 0
    cdna = os.path.join(ROOT, "ginkgo_amd", "lib", "libgko_cdna4.so")
    if not os.path.exists(cdna) or not os.path.exists(os.path.join(REFB, "lib", "libginkgo.so")):
        print("[build_dropin] build libgko_cdna4.so and oracle/_ref first")
        return 1
    os.makedirs(OUT, exist_ok=True)
    inc = [f"-I{REFB}/include", f"-I{ref}/include", f"-I{ref}", f"-I{ROOT}/include"]
    flags = ["-std=c++17", "-O2", "-fPIC", "-w", "-pthread"]
    # 1. + 2. the product's own build (ginkgo_amd/gko_binding/build.py -> ginkgo_amd/lib/); the test
    #    programs below find the library next to them, as a Ginkgo build tree would have it
    p = subprocess.run([sys.executable, os.path.join(ROOT, "ginkgo_amd", "gko_binding", "build.py")],
                       env=dict(os.environ, GKO_BUILD_DIR=REFB, GKO_REFERENCE_DIR=ref))
    if p.returncode != 0:
        return 1
    lib = os.path.join(OUT, "libginkgo_hip.so")
    shutil.copy(os.path.join(ROOT, "ginkgo_amd", "lib", "libginkgo_hip.so"), lib)
    # 3. acceptance programs; the dropin directory comes FIRST in the rpath so
    #    that libginkgo.so's NEEDED libginkgo_hip.so resolves to the shim
    link = [f"-L{OUT}", f"-L{REFB}/lib", "-lginkgo", "-lginkgo_omp", "-lginkgo_reference",
            "-lginkgo_hip", "-lginkgo_cuda", "-lginkgo_dpcpp", "-lginkgo_device", "-fopenmp",
            "-Wl,-rpath,$ORIGIN:$ORIGIN/../lib:$ORIGIN/../../../ginkgo_amd/lib"]
    run(["g++"] + flags + inc + [os.path.join(ROOT, "tests", "dropin", "dropin_test.cpp"), "-o",
                                 os.path.join(OUT, "dropin_test")] + link +
        [f"-L{os.path.dirname(cdna)}", "-lgko_cdna4"])     # gkoc_tune_set: the fusion switch
    run(["g++"] + flags + inc + [os.path.join(ROOT, "tests", "dropin", "dropin_bench.cpp"), "-o",
                                 os.path.join(OUT, "dropin_bench")] + link +
        [f"-L{os.path.dirname(cdna)}", "-lgko_cdna4"])
    run(["g++"] + flags + inc + [os.path.join(ROOT, "tests", "dropin", "round5_bench.cpp"), "-o",
                                 os.path.join(OUT, "round5_bench")] + link +
        [f"-L{os.path.dirname(cdna)}", "-lgko_cdna4"])
    run(["g++"] + flags + inc + [os.path.join(ROOT, "tests", "dropin", "arena_roles_test.cpp"), "-o",
                                 os.path.join(OUT, "arena_roles_test")] + link +
        [f"-L{os.path.dirname(cdna)}", "-lgko_cdna4"])
    # the GINKGO_MIXED_PRECISION flavor of the core (oracle/build_ref_mixed.py) on the same shim
    mixed = os.path.join(REFB, "mixed")
    if os.path.exists(os.path.join(mixed, "lib", "libginkgo.so")):
        minc = [f"-I{mixed}/include", f"-I{ref}/include", f"-I{ref}", f"-I{ROOT}/include"]
        mlink = [f"-L{OUT}", f"-L{mixed}/lib", f"-L{REFB}/lib", "-lginkgo", "-lginkgo_omp", "-lginkgo_reference",
                 "-lginkgo_hip", "-lginkgo_cuda", "-lginkgo_dpcpp", "-lginkgo_device", "-fopenmp",
                 "-Wl,-rpath,$ORIGIN:$ORIGIN/../mixed/lib:$ORIGIN/../lib:$ORIGIN/../../../ginkgo_amd/lib"]
        run(["g++"] + flags + minc + [os.path.join(ROOT, "tests", "dropin", "mixed_test.cpp"), "-o",
                                      os.path.join(OUT, "mixed_test")] + mlink)
    ex_src = os.path.join(ref, "examples", "preconditioned-solver", "preconditioned-solver.cpp")
    wrap = os.path.join(OUT, "ginkgo_all.hpp")
    # <ginkgo/ginkgo.hpp> is CMake-generated: provide an umbrella of the public headers
    hdrs = sorted(glob.glob(os.path.join(ref, "include", "ginkgo", "core", "**", "*.hpp"), recursive=True))
    os.makedirs(os.path.join(OUT, "include", "ginkgo"), exist_ok=True)
    with open(os.path.join(OUT, "include", "ginkgo", "ginkgo.hpp"), "w") as f:
        f.write("// umbrella header generated by oracle/build_dropin.py\n#pragma once\n")
        f.write("#include <ginkgo/config.hpp>\n")
        for h in hdrs:
            rel = os.path.relpath(h, os.path.join(ref, "include"))
            if "/distributed/" in rel or rel.endswith("mpi.hpp") or "/synthesizer/" in rel:
                continue
            f.write(f"#include <{rel}>\n")
    run(["g++"] + flags + [f"-I{OUT}/include"] + inc + [ex_src, "-o",
                                                        os.path.join(OUT, "preconditioned-solver")] + link)
    data_dst = os.path.join(OUT, "data")
    os.makedirs(data_dst, exist_ok=True)
    for fn in glob.glob(os.path.join(ref, "examples", "preconditioned-solver", "data", "*.mtx")):
        shutil.copy(fn, data_dst)
    print(f"[build_dropin] built {lib}, dropin_test, dropin_bench, preconditioned-solver")
    return 0


if __name__ == "__main__":
    sys.exit(main())
