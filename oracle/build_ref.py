#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- compile the unmodified reference into oracle/_ref.

Builds Ginkgo 1.12.0's core + `reference` + `omp` kernel libraries (and the
reference's own "not compiled" stubs for cuda/dpcpp/hip,
core/device_hooks/*_hooks.cpp) straight from the sources where they lie under
/root/reference, with plain g++ -- the reference's CMake is NOT run.  The source
list (oracle/ref_sources.tsv: "<library>\t<path relative to the reference>") is
the set of translation units a CPU-only configuration of the reference
compiles; oracle/ref_config.hpp stands in for the generated <ginkgo/config.hpp>.

Outputs go only to oracle/_ref/ (git-ignored; travels to the GPU box with the
snapshot).  Nothing is copied from the reference into the repository.

Used for: (1) pinning oracle/gko_oracle.c against the reference's own
ReferenceExecutor kernels, (2) the `cpu_baseline.kind == "reference"` leg of
bench.py (OmpExecutor), (3) demonstrating the drop-in: the real libginkgo.so
running on top of our libginkgo_hip.so replacement.

Usage: python oracle/build_ref.py [-j N] [--ref /root/reference]
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

LIB_FLAGS = {
    "core": ["-Dginkgo_EXPORTS"],
    "device": ["-Dginkgo_device_EXPORTS"],
    "reference": ["-Dginkgo_reference_EXPORTS", "-DGKO_COMPILING_REFERENCE",
                  "-DGKO_DEVICE_NAMESPACE=reference"],
    "omp": ["-Dginkgo_omp_EXPORTS", "-DGKO_COMPILING_OMP",
            "-DGKO_DEVICE_NAMESPACE=omp", "-fopenmp"],
    "cuda": ["-Dginkgo_cuda_EXPORTS"],
    "dpcpp": ["-Dginkgo_dpcpp_EXPORTS"],
    "hip": ["-Dginkgo_hip_EXPORTS"],
}
# link order: later entries may depend on earlier ones
LINK_ORDER = ["device", "reference", "omp", "cuda", "dpcpp", "hip", "core"]
LIB_NAME = {
    "core": "libginkgo.so", "device": "libginkgo_device.so",
    "reference": "libginkgo_reference.so", "omp": "libginkgo_omp.so",
    "cuda": "libginkgo_cuda.so", "dpcpp": "libginkgo_dpcpp.so",
    "hip": "libginkgo_hip.so",
}
LIB_DEPS = {
    "device": [], "reference": ["device"], "omp": ["device"],
    "cuda": ["device"], "dpcpp": ["device"], "hip": ["device"],
    "core": ["omp", "cuda", "reference", "hip", "dpcpp", "device"],
}


def read_sources():
    libs = {}
    with open(os.path.join(HERE, "ref_sources.tsv")) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            lib, path = line.split("\t")
            libs.setdefault(lib, []).append(path)
    return libs


def compile_one(args):
    cxx, ref, lib, rel, flags = args
    src = os.path.join(ref, rel)
    obj = os.path.join(OUT, "obj", lib, rel.replace("/", "__") + ".o")
    os.makedirs(os.path.dirname(obj), exist_ok=True)
    if os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src):
        return obj, 0.0, ""
    inc = [f"-I{OUT}/include", f"-I{ref}/include", f"-I{ref}"]
    if lib == "omp":
        inc.insert(0, f"-I{ref}/omp")
    if lib == "reference":
        inc.insert(0, f"-I{ref}/reference")
    cmd = [cxx, "-std=c++17", "-O3", "-DNDEBUG", "-fPIC", "-w"] + flags + inc + \
        ["-c", src, "-o", obj]
    t0 = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        return obj, time.time() - t0, " ".join(cmd) + "\n" + p.stderr[-4000:]
    return obj, time.time() - t0, ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=os.cpu_count() or 4)
    ap.add_argument("--ref", default=os.environ.get("GKO_REFERENCE_DIR",
                                                    "/root/reference"))
    ap.add_argument("--cxx", default=os.environ.get("CXX", "g++"))
    ap.add_argument("--only", default="", help="comma list of libs to build")
    a = ap.parse_args()
    if not os.path.isdir(os.path.join(a.ref, "core")):
        print(f"[build_ref] reference not found at {a.ref}; keeping prebuilt "
              f"oracle/_ref as is")
        return 0
    os.makedirs(os.path.join(OUT, "include", "ginkgo"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "lib"), exist_ok=True)
    cfg_src = open(os.path.join(HERE, "ref_config.hpp")).read()
    cfg_dst = os.path.join(OUT, "include", "ginkgo", "config.hpp")
    if not os.path.exists(cfg_dst) or open(cfg_dst).read() != cfg_src:
        open(cfg_dst, "w").write(cfg_src)
    libs = read_sources()
    only = [x for x in a.only.split(",") if x]
    jobs = []
    for lib, srcs in libs.items():
        if only and lib not in only:
            continue
        for rel in srcs:
            jobs.append((a.cxx, a.ref, lib, rel, LIB_FLAGS[lib]))
    # big translation units first => better packing
    jobs.sort(key=lambda j: -os.path.getsize(os.path.join(j[1], j[3])))
    objs = {}
    t0 = time.time()
    failed = False
    with cf.ThreadPoolExecutor(max_workers=a.j) as ex:
        for n, (job, res) in enumerate(zip(jobs, ex.map(compile_one, jobs))):
            obj, dt, err = res
            if err:
                print(f"[build_ref] FAILED {job[3]}\n{err}", file=sys.stderr)
                failed = True
            objs.setdefault(job[2], []).append(obj)
            if dt > 0 and (n % 20 == 0):
                print(f"[build_ref] {n + 1}/{len(jobs)} {job[3]} ({dt:.0f}s, "
                      f"elapsed {time.time() - t0:.0f}s)", flush=True)
    if failed:
        return 1
    for lib in LINK_ORDER:
        if lib not in objs:
            continue
        out = os.path.join(OUT, "lib", LIB_NAME[lib])
        newest = max(os.path.getmtime(o) for o in objs[lib])
        if os.path.exists(out) and os.path.getmtime(out) >= newest:
            continue
        cmd = [a.cxx, "-shared", "-fPIC", "-s", "-o", out,
               f"-Wl,-soname,{LIB_NAME[lib]}", "-Wl,-rpath,$ORIGIN"] + \
            sorted(objs[lib]) + [f"-L{OUT}/lib"] + \
            ["-l" + LIB_NAME[d][3:-3] for d in LIB_DEPS[lib]]
        if lib in ("omp", "core"):
            cmd.append("-fopenmp")
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            print(f"[build_ref] link {lib} failed:\n{p.stderr[-4000:]}",
                  file=sys.stderr)
            return 1
        print(f"[build_ref] linked {out}", flush=True)
    print(f"[build_ref] done in {time.time() - t0:.0f}s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
