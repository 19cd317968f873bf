#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- the reference once more with GINKGO_MIXED_PRECISION defined
(CMake's -DGINKGO_MIXED_PRECISION=ON, include/ginkgo/config.hpp.in:53), into oracle/_ref/mixed/:
Csr / Ell / Dense then dispatch apply() over every (matrix, input, output) value-type triple
(include/ginkgo/core/base/precision_dispatch.hpp, core/base/mixed_precision_types.hpp) instead of
converting the vectors first.  tests/dropin/mixed_test.cpp runs against this flavor and the SAME
libginkgo_hip.so as every other flavor (the shim defines the mixed instantiations
unconditionally).

Only the translation units that read the switch are compiled again (`grep -rlE
"mixed_precision_dispatch|MIXED_VALUE|GINKGO_MIXED_PRECISION" core reference omp common/unified`);
every other object is the one oracle/build_ref.py built.  Outputs: oracle/_ref/mixed/{include/
ginkgo/config.hpp, obj/, lib/libginkgo{,_reference,_omp,_cuda,_dpcpp,_hip}.so}."""
import concurrent.futures as cf
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from build_ref import LIB_DEPS, LIB_FLAGS, LIB_NAME, LINK_ORDER, read_sources  # noqa: E402

REFB = os.path.join(HERE, "_ref")
OUT = os.path.join(REFB, "mixed")

AFFECTED = {
    "core": ["core/matrix/csr.cpp", "core/matrix/dense.cpp", "core/matrix/ell.cpp",
             "core/matrix/sparsity_csr.cpp"],
    "reference": ["reference/matrix/csr_kernels.cpp", "reference/matrix/dense_kernels.cpp",
                  "reference/matrix/ell_kernels.cpp", "reference/matrix/sparsity_csr_kernels.cpp"],
    "omp": ["omp/matrix/csr_kernels.cpp", "omp/matrix/ell_kernels.cpp",
            "omp/matrix/sparsity_csr_kernels.cpp", "common/unified/matrix/dense_kernels.instantiate.cpp"],
    "cuda": ["core/device_hooks/cuda_hooks.cpp"],
    "dpcpp": ["core/device_hooks/dpcpp_hooks.cpp"],
    "hip": ["core/device_hooks/hip_hooks.cpp"],
}


def obj_name(base, lib, rel):
    return os.path.join(base, "obj", lib, rel.replace("/", "__") + ".o")


def compile_one(args):
    ref, lib, rel = args
    src = os.path.join(ref, rel)
    obj = obj_name(OUT, lib, rel)
    os.makedirs(os.path.dirname(obj), exist_ok=True)
    if os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src):
        return obj, ""
    inc = [f"-I{OUT}/include", f"-I{ref}/include", f"-I{ref}"]
    if lib in ("omp", "reference"):
        inc.insert(0, f"-I{ref}/{lib}")
    cmd = ["g++", "-std=c++17", "-O3", "-DNDEBUG", "-fPIC", "-w"] + LIB_FLAGS[lib] + inc + \
        ["-c", src, "-o", obj]
    p = subprocess.run(cmd, capture_output=True, text=True)
    return obj, ("" if p.returncode == 0 else " ".join(cmd) + "\n" + p.stderr[-4000:])


def main():
    ref = os.environ.get("GKO_REFERENCE_DIR", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "core")):
        print("[build_ref_mixed] reference not found; keeping prebuilt outputs")
        return 0
    if not os.path.exists(os.path.join(REFB, "lib", "libginkgo.so")):
        print("[build_ref_mixed] run build_ref.py first")
        return 1
    os.makedirs(os.path.join(OUT, "include", "ginkgo"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "lib"), exist_ok=True)
    cfg = open(os.path.join(HERE, "ref_config.hpp")).read()
    assert "GINKGO_MIXED_PRECISION" not in cfg
    mark = "#define GKO_VERBOSE_LEVEL 1"
    assert mark in cfg
    cfg = cfg.replace(mark, "#define GINKGO_MIXED_PRECISION\n" + mark)
    dst = os.path.join(OUT, "include", "ginkgo", "config.hpp")
    if not os.path.exists(dst) or open(dst).read() != cfg:
        open(dst, "w").write(cfg)
    libs = read_sources()
    jobs = []
    for lib, rels in AFFECTED.items():
        for rel in rels:
            assert rel in libs[lib], f"{rel} is not a source of {lib}"
            jobs.append((ref, lib, rel))
    jobs.sort(key=lambda j: -os.path.getsize(os.path.join(j[0], j[2])))
    t0 = time.time()
    failed = False
    with cf.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
        for job, (obj, err) in zip(jobs, ex.map(compile_one, jobs)):
            if err:
                print(f"[build_ref_mixed] FAILED {job[2]}\n{err}", file=sys.stderr)
                failed = True
    if failed:
        return 1
    for lib in LINK_ORDER:
        if lib == "device":
            continue        # untouched: found through the rpath
        objs = []
        for rel in libs[lib]:
            mine = obj_name(OUT, lib, rel)
            objs.append(mine if rel in AFFECTED.get(lib, []) else obj_name(REFB, lib, rel))
            if not os.path.exists(objs[-1]):
                print(f"[build_ref_mixed] missing object {objs[-1]}: run build_ref.py", file=sys.stderr)
                return 1
        out = os.path.join(OUT, "lib", LIB_NAME[lib])
        if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(o) for o in objs):
            continue
        cmd = ["g++", "-shared", "-fPIC", "-s", "-o", out, f"-Wl,-soname,{LIB_NAME[lib]}",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../../lib"] + sorted(objs) + \
            [f"-L{OUT}/lib", f"-L{REFB}/lib"] + ["-l" + LIB_NAME[d][3:-3] for d in LIB_DEPS[lib]]
        if lib in ("omp", "core"):
            cmd.append("-fopenmp")
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            print(f"[build_ref_mixed] link {lib} failed:\n{p.stderr[-4000:]}", file=sys.stderr)
            return 1
    print(f"[build_ref_mixed] built {OUT}/lib in {time.time() - t0:.0f}s")
    # the C-callable front of this flavor (oracle/ref_shim_mixed.cpp) for the oracle's pinning tests
    src = os.path.join(HERE, "ref_shim_mixed.cpp")
    shim = os.path.join(OUT, "libgko_ref_shim_mixed.so")
    if not os.path.exists(shim) or os.path.getmtime(shim) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(OUT, "lib", "libginkgo.so"))):
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-w", "-fopenmp", f"-I{OUT}/include",
               f"-I{ref}/include", f"-I{ref}", src, "-o", shim, f"-L{OUT}/lib", f"-L{REFB}/lib", "-lginkgo",
               "-lginkgo_omp", "-lginkgo_reference", "-lginkgo_hip", "-lginkgo_cuda", "-lginkgo_dpcpp",
               "-lginkgo_device", "-Wl,-rpath,$ORIGIN/lib:$ORIGIN/../lib"]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            print(p.stderr[-6000:], file=sys.stderr)
            return 1
        print(f"[build_ref_mixed] built {shim}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
