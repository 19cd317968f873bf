#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- builds oracle/_ref/libgko_ref_shim.so (oracle/ref_shim.cpp)
against the reference built by oracle/build_ref.py.  Needs /root/reference for
the headers; on the GPU box the prebuilt .so is used as is."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")


def main():
    ref = os.environ.get("GKO_REFERENCE_DIR", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "include", "ginkgo")):
        print("[build_shim] reference headers not found; keeping prebuilt shim")
        return 0
    src = os.path.join(HERE, "ref_shim.cpp")
    out = os.path.join(OUT, "libgko_ref_shim.so")
    lib = os.path.join(OUT, "lib", "libginkgo.so")
    if not os.path.exists(lib):
        print("[build_shim] oracle/_ref/lib/libginkgo.so missing: run build_ref.py")
        return 1
    if os.path.exists(out) and os.path.getmtime(out) >= max(
            os.path.getmtime(src), os.path.getmtime(lib)):
        return 0
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-w", "-fopenmp",
           f"-I{OUT}/include", f"-I{ref}/include", f"-I{ref}", src, "-o", out,
           f"-L{OUT}/lib", "-lginkgo", "-lginkgo_omp", "-lginkgo_reference",
           "-lginkgo_device", "-Wl,-rpath,$ORIGIN/lib"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        print(p.stderr[-6000:], file=sys.stderr)
        return 1
    print(f"[build_shim] built {out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
