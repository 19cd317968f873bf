#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- the reference's core library once more, with GINKGO_BUILD_MPI=1,
into oracle/_ref/mpi/ (MPICH 3.3.2 of the image, /opt/conda): gko::experimental::distributed::
{Matrix, Vector, RowGatherer, ...} and the distributed solver instantiations, unmodified, so that
Ginkgo's own distributed classes and its examples/distributed-solver run on this backend
(tests/test_mpi_dropin_gpu.py).  Only libginkgo.so is rebuilt (the kernel
libraries of oracle/build_ref.py do not depend on the MPI switch); the extra translation units
are the ones core/CMakeLists.txt:13-15,145-161 adds for GINKGO_BUILD_MPI.

Outputs: oracle/_ref/mpi/{include/ginkgo/config.hpp, obj/, lib/libginkgo.so, lib/libmpi.so.12 ...}.
The three MPICH runtime libraries are copied next to it so that no conda directory has to be on
the library path (conda ships its own libgomp / libstdc++).  MPICH here is NOT GPU-aware:
Ginkgo stages device buffers through the host by itself (mpi::requires_host_buffer)."""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REFB = os.path.join(HERE, "_ref")
OUT = os.path.join(REFB, "mpi")
MPI_ROOT = os.environ.get("GKO_MPI_ROOT", "/opt/conda")

EXTRA = ["core/config/schwarz_config.cpp", "core/distributed/assembly.cpp",
         "core/distributed/collective_communicator.cpp", "core/distributed/dense_communicator.cpp",
         "core/distributed/matrix.cpp", "core/distributed/neighborhood_communicator.cpp",
         "core/distributed/partition_helpers.cpp", "core/distributed/preconditioner/schwarz.cpp",
         "core/distributed/row_gatherer.cpp", "core/distributed/vector.cpp",
         "core/distributed/vector_cache.cpp", "core/mpi/exception.cpp"]


def compile_one(args):
    ref, rel = args
    src = os.path.join(ref, rel)
    obj = os.path.join(OUT, "obj", rel.replace("/", "__") + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src):
        return obj, ""
    cmd = ["g++", "-std=c++17", "-O2", "-DNDEBUG", "-fPIC", "-w", "-Dginkgo_EXPORTS",
           f"-I{OUT}/include", f"-I{ref}/include", f"-I{ref}", "-idirafter", f"{MPI_ROOT}/include",
           "-c", src, "-o", obj]
    p = subprocess.run(cmd, capture_output=True, text=True)
    return obj, ("" if p.returncode == 0 else " ".join(cmd) + "\n" + p.stderr[-4000:])


def main():
    ref = os.environ.get("GKO_REFERENCE_DIR", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "core")):
        print("[build_ref_mpi] reference not found; keeping prebuilt outputs")
        return 0
    if not os.path.exists(os.path.join(MPI_ROOT, "include", "mpi.h")):
        print(f"[build_ref_mpi] no MPI under {MPI_ROOT}: skipped")
        return 0
    if not os.path.exists(os.path.join(REFB, "lib", "libginkgo_reference.so")):
        print("[build_ref_mpi] run build_ref.py first")
        return 1
    for d in ("include/ginkgo", "obj", "lib"):
        os.makedirs(os.path.join(OUT, d), exist_ok=True)
    cfg = open(os.path.join(HERE, "ref_config.hpp")).read()
    assert "#define GINKGO_BUILD_MPI 0" in cfg
    cfg = cfg.replace("#define GINKGO_BUILD_MPI 0", "#define GINKGO_BUILD_MPI 1")
    dst = os.path.join(OUT, "include", "ginkgo", "config.hpp")
    if not os.path.exists(dst) or open(dst).read() != cfg:
        open(dst, "w").write(cfg)
    srcs = [l.split("\t")[1].strip() for l in open(os.path.join(HERE, "ref_sources.tsv"))
            if l.startswith("core\t")] + EXTRA
    srcs.sort(key=lambda r: -os.path.getsize(os.path.join(ref, r)))
    t0 = time.time()
    objs, failed = [], False
    with cf.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for rel, (obj, err) in zip(srcs, ex.map(compile_one, [(ref, r) for r in srcs])):
            if err:
                print(f"[build_ref_mpi] FAILED {rel}\n{err}", file=sys.stderr)
                failed = True
            objs.append(obj)
    if failed:
        return 1
    for so in ("libmpi.so.12", "libgfortran.so.4", "libquadmath.so.0"):
        if not os.path.exists(os.path.join(OUT, "lib", so)):
            shutil.copy(os.path.realpath(os.path.join(MPI_ROOT, "lib", so)), os.path.join(OUT, "lib", so))
    out = os.path.join(OUT, "lib", "libginkgo.so")
    cmd = ["g++", "-shared", "-fPIC", "-s", "-o", out, "-Wl,-soname,libginkgo.so",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../../lib"] + sorted(objs) + \
        [f"-L{REFB}/lib", "-lginkgo_omp", "-lginkgo_cuda", "-lginkgo_reference", "-lginkgo_hip",
         "-lginkgo_dpcpp", "-lginkgo_device", os.path.join(OUT, "lib", "libmpi.so.12"), "-fopenmp"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        print(f"[build_ref_mpi] link failed:\n{p.stderr[-4000:]}", file=sys.stderr)
        return 1
    print(f"[build_ref_mpi] built {out} in {time.time() - t0:.0f}s")
    return build_gpu_aware_flavor(ref, objs)


def build_gpu_aware_flavor(ref, objs):
    """The same core once more with GINKGO_HAVE_GPU_AWARE_MPI 1 - what Ginkgo's CMake sets for
    GINKGO_FORCE_GPU_AWARE_MPI (CMakeLists.txt:185, 410-417) - into oracle/_ref/mpi_ga/:
    mpi::requires_host_buffer is then false (core/base/mpi.cpp:65-70) and the distributed classes
    hand DEVICE pointers to MPI, which libgkoc_mpi_rccl.so (ginkgo_amd/gko_binding/mpi_rccl.cpp)
    routes over RCCL.  The switch is read in exactly one translation unit of the core
    (core/base/mpi.cpp:69 through mpi.hpp:42-49 is_gpu_aware; `grep -rn is_gpu_aware` over core/,
    include/), so only that one is compiled again; every other object is the one built above."""
    ga = os.path.join(REFB, "mpi_ga")
    for d in ("include/ginkgo", "obj", "lib"):
        os.makedirs(os.path.join(ga, d), exist_ok=True)
    cfg = open(os.path.join(OUT, "include", "ginkgo", "config.hpp")).read()
    assert "#define GINKGO_HAVE_GPU_AWARE_MPI 0" in cfg
    cfg = cfg.replace("#define GINKGO_HAVE_GPU_AWARE_MPI 0", "#define GINKGO_HAVE_GPU_AWARE_MPI 1")
    dst = os.path.join(ga, "include", "ginkgo", "config.hpp")
    if not os.path.exists(dst) or open(dst).read() != cfg:
        open(dst, "w").write(cfg)
    rel = "core/base/mpi.cpp"
    obj = os.path.join(ga, "obj", rel.replace("/", "__") + ".o")
    if not os.path.exists(obj) or os.path.getmtime(obj) < os.path.getmtime(dst):
        cmd = ["g++", "-std=c++17", "-O2", "-DNDEBUG", "-fPIC", "-w", "-Dginkgo_EXPORTS",
               f"-I{ga}/include", f"-I{ref}/include", f"-I{ref}", "-idirafter", f"{MPI_ROOT}/include",
               "-c", os.path.join(ref, rel), "-o", obj]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode:
            print(f"[build_ref_mpi] gpu-aware flavor: {rel} failed\n{p.stderr[-4000:]}", file=sys.stderr)
            return 1
    plain = os.path.join(OUT, "obj", rel.replace("/", "__") + ".o")
    use = [o for o in objs if os.path.abspath(o) != os.path.abspath(plain)] + [obj]
    assert len(use) == len(objs), "core/base/mpi.cpp is not among the core objects"
    for so in ("libmpi.so.12", "libgfortran.so.4", "libquadmath.so.0"):
        if not os.path.exists(os.path.join(ga, "lib", so)):
            shutil.copy(os.path.join(OUT, "lib", so), os.path.join(ga, "lib", so))
    out = os.path.join(ga, "lib", "libginkgo.so")
    cmd = ["g++", "-shared", "-fPIC", "-s", "-o", out, "-Wl,-soname,libginkgo.so",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../../lib"] + sorted(use) + \
        [f"-L{REFB}/lib", "-lginkgo_omp", "-lginkgo_cuda", "-lginkgo_reference", "-lginkgo_hip",
         "-lginkgo_dpcpp", "-lginkgo_device", os.path.join(ga, "lib", "libmpi.so.12"), "-fopenmp"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        print(f"[build_ref_mpi] gpu-aware flavor: link failed:\n{p.stderr[-4000:]}", file=sys.stderr)
        return 1
    print(f"[build_ref_mpi] built {out} (GINKGO_HAVE_GPU_AWARE_MPI 1)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
