#!/usr/bin/env python3
"""Builds the Ginkgo-facing libraries of the product from this directory, against a Ginkgo source
tree (this is a Ginkgo backend: it is compiled against Ginkgo's headers like Ginkgo's own hip/):

  ginkgo_amd/lib/libginkgo_hip.so     the link-compatible replacement for Ginkgo's HIP module:
        gko_binding/*.cpp (strong gko::kernels::hip::* / gko::HipExecutor symbols forwarding to
        the C ABI of libgko_cdna4.so) + Ginkgo's own stub translation unit
        core/device_hooks/hip_hooks.cpp with every symbol WEAKENED (objcopy --weaken: whatever
        this backend does not define keeps throwing gko::NotCompiled, exactly as in a Ginkgo build
        without the module) + devices/hip/executor.cpp
  ginkgo_amd/lib/libgkoc_mpi_rccl.so  the GPU-aware-MPI layer (mpi_rccl.cpp) for a Ginkgo built
        with GINKGO_BUILD_MPI + GINKGO_FORCE_GPU_AWARE_MPI on an MPI that is not GPU-aware:
        device buffers of Ginkgo's own MPI calls go over RCCL (only if an mpi.h is found)

Inputs (environment):
  GKO_REFERENCE_DIR   Ginkgo source tree (default /root/reference)
  GKO_BUILD_DIR       a build of that tree: include/ginkgo/config.hpp and lib/libginkgo_device.so
                      (required; this repository's test infrastructure builds the unmodified
                      reference with plain g++ and passes its output directory)
  GKO_MPI_ROOT        MPI installation (default /opt/conda)
Intermediate objects: ginkgo_amd/gko_binding/build/ (not tracked)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "ginkgo_amd", "lib")
OBJ = os.path.join(HERE, "build")


def run(cmd):
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        print(" ".join(cmd), file=sys.stderr)
        print(p.stderr[-8000:], file=sys.stderr)
        sys.exit(1)


def newer(out, *deps):
    return os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in deps)


def main():
    ref = os.environ.get("GKO_REFERENCE_DIR", "/root/reference")
    bld = os.environ.get("GKO_BUILD_DIR", "")
    mpi_root = os.environ.get("GKO_MPI_ROOT", "/opt/conda")
    if not os.path.isdir(os.path.join(ref, "include", "ginkgo")):
        print("[gko_binding/build] no Ginkgo source tree; keeping the prebuilt libraries")
        return 0
    cdna = os.path.join(LIB, "libgko_cdna4.so")
    if not os.path.exists(cdna):
        print("[gko_binding/build] build libgko_cdna4.so first (make -C ginkgo_amd/csrc)")
        return 1
    if not bld:
        print("[gko_binding/build] set GKO_BUILD_DIR to a Ginkgo build directory")
        return 1
    if not os.path.exists(os.path.join(bld, "include", "ginkgo", "config.hpp")) or \
            not os.path.exists(os.path.join(bld, "lib", "libginkgo_device.so")):
        print(f"[gko_binding/build] {bld} holds no Ginkgo build (config.hpp, libginkgo_device.so)")
        return 1
    os.makedirs(OBJ, exist_ok=True)
    inc = [f"-I{bld}/include", f"-I{ref}/include", f"-I{ref}", f"-I{ROOT}/include"]
    flags = ["-std=c++17", "-O2", "-fPIC", "-w"]
    hdrs = [os.path.join(ROOT, "include", "gko_cdna4.h"), os.path.join(HERE, "shim_common.hpp")]
    objs = []
    for src in sorted(glob.glob(os.path.join(HERE, "*.cpp"))):
        if os.path.basename(src) == "mpi_rccl.cpp":
            continue
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        if not newer(obj, src, *hdrs):
            run(["g++"] + flags + inc + ["-c", src, "-o", obj])
        objs.append(obj)
    # Ginkgo's own stub translation unit (weakened) and the host side of HipExecutor
    extra = []
    for rel in ("core/device_hooks/hip_hooks.cpp", "devices/hip/executor.cpp"):
        obj = os.path.join(OBJ, rel.replace("/", "__") + ".o")
        if not newer(obj, os.path.join(ref, rel)):
            # the stubs of a GINKGO_MIXED_PRECISION core as well (a superset: one library serves
            # both kinds of core; mixed.cpp defines the hot-path triples, the rest stays NotCompiled)
            run(["g++", "-std=c++17", "-O2", "-DNDEBUG", "-DGINKGO_MIXED_PRECISION", "-fPIC", "-w"] + inc +
                ["-c", os.path.join(ref, rel), "-o", obj])
        extra.append(obj)
    weak = os.path.join(OBJ, "hip_hooks_weak.o")
    run(["objcopy", "--weaken", extra[0], weak])
    lib = os.path.join(LIB, "libginkgo_hip.so")
    run(["g++", "-shared", "-fPIC", "-o", lib, "-Wl,-soname,libginkgo_hip.so"] + objs + [weak, extra[1]] +
        [f"-L{bld}/lib", "-lginkgo_device", f"-L{LIB}", "-lgko_cdna4",
         "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../lib",
         "-Wl,-rpath,$ORIGIN/../../../ginkgo_amd/lib", "-Wl,-rpath," + os.path.join(bld, "lib")])
    built = [lib]
    if os.path.exists(os.path.join(mpi_root, "include", "mpi.h")):
        out = os.path.join(LIB, "libgkoc_mpi_rccl.so")
        src = os.path.join(HERE, "mpi_rccl.cpp")
        libmpi = os.path.join(mpi_root, "lib", "libmpi.so")
        run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-w", f"-I{ROOT}/include", "-idirafter",
             f"{mpi_root}/include", src, "-o", out, "-Wl,-soname,libgkoc_mpi_rccl.so", f"-L{LIB}",
             "-lgko_cdna4", libmpi, "-Wl,-rpath,$ORIGIN"])
        built.append(out)
    print("[gko_binding/build] built " + ", ".join(os.path.relpath(b, ROOT) for b in built))
    return 0


if __name__ == "__main__":
    sys.exit(main())
