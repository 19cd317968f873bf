#!/usr/bin/env python3
"""Generates tests/golden/jacobi_types.npz with the UNMODIFIED reference (oracle/_ref, built by
oracle/build_ref.py) through tests/golden/jacobi_types_ref.cpp: block-Jacobi with fixed reduced,
autodetected and block-wise storage precisions for float, complex<float> and complex<double> on
gko::ReferenceExecutor - the blocks found, every block's precision and condition number, and
M b, M^T b, M^H b for 3 right-hand sides.

    python tests/golden/make_jacobi_types_golden.py          # write the fixture
    python tests/golden/make_jacobi_types_golden.py --check  # the fixture == the live reference

The fixture travels to the GPU box, where tests/test_jacobi_types_gpu.py puts the same inputs through
the C ABI (gkoc_jacobi_*_adaptive_{f32,c64,c128}_i32)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFB = os.path.join(ROOT, "oracle", "_ref")
SHIM = os.path.join(REFB, "libgko_ref_jacobi_types.so")
OUT = os.path.join(HERE, "jacobi_types.npz")

VT = {0: ("f32", np.float32), 1: ("c64", np.complex64), 2: ("c128", np.complex128)}
SIZES = {0: (8, 32), 1: (13,), 2: (8, 32)}
MIX = [0x01, 0xff, 0x20, 0x00, 0xff, 0x11, 0x02]
REQUESTS = [("p01", [0x01], 1e-1), ("p02", [0x02], 1e-1), ("p10", [0x10], 1e-1), ("p11", [0x11], 1e-1),
            ("p20", [0x20], 1e-1), ("auto1", [0xff], 1e-1), ("auto3", [0xff], 1e-3), ("mix", MIX, 1e-2)]
N = 301
NRHS = 3


def build_shim():
    ref = os.environ.get("GKO_REFERENCE_DIR", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "include", "ginkgo")):
        raise SystemExit("needs the reference sources (headers) and oracle/_ref")
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-w", f"-I{REFB}/include", f"-I{ref}/include",
           f"-I{ref}", os.path.join(HERE, "jacobi_types_ref.cpp"), "-o", SHIM, f"-L{REFB}/lib", "-lginkgo",
           "-lginkgo_omp", "-lginkgo_reference", "-lginkgo_hip", "-lginkgo_cuda", "-lginkgo_dpcpp",
           "-lginkgo_device", "-fopenmp", "-Wl,-rpath,$ORIGIN/lib"]
    subprocess.run(cmd, check=True)


def inputs(vt):
    """rows of a band matrix whose diagonal blocks go from well conditioned to nearly singular, a far
    coupling outside the blocks; complex types get imaginary parts"""
    dt = VT[vt][1]
    cplx = np.issubdtype(dt, np.complexfloating)
    shift = [2.0, 0.5, 0.01, 1e-4]
    rows, cols, vals = [], [], []
    for i in range(N):
        ent = []
        if i >= 41:
            ent.append((i - 41, -0.05 + 0.02j))
        if i > 0:
            ent.append((i - 1, -1.0 + 0.3j))
        ent.append((i, 2.0 + shift[(i // 75) % 4] + 0.2j))
        if i + 1 < N:
            ent.append((i + 1, -1.0 - 0.25j))
        if i + 41 < N:
            ent.append((i + 41, -0.1 + 0.05j))
        for c, v in ent:
            rows.append(i)
            cols.append(c)
            vals.append(v if cplx else v.real)
    rp = np.zeros(N + 1, np.int32)
    np.add.at(rp, np.asarray(rows) + 1, 1)
    rp = np.cumsum(rp).astype(np.int32)
    k = np.arange(N * NRHS, dtype=np.float64).reshape(N, NRHS)
    b = np.cos(0.11 * k) + (1j * np.sin(0.07 * k) if cplx else 0.0)
    return rp, np.asarray(cols, np.int32), np.asarray(vals).astype(dt), np.ascontiguousarray(b.astype(dt))


def reference(lib, vt, rp, ci, vals, b, max_bs, req, accuracy):
    dt = VT[vt][1]
    x, xt, xh = (np.zeros((N, NRHS), dt) for _ in range(3))
    bp = np.zeros(N + 1, np.int32)
    prec = np.zeros(N, np.uint8)
    cond = np.zeros(N, np.float64)
    rq = np.asarray(req, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
    nb = lib.ref_jacobi_types(C.c_int(vt), C.c_int64(N), p(rp), p(ci), p(vals), C.c_uint32(max_bs), p(rq),
                              C.c_int64(len(rq)), C.c_double(accuracy), C.c_int64(NRHS), p(b), p(x), p(xt),
                              p(xh), p(bp), p(prec), p(cond))
    assert nb > 0, nb
    return {"block_ptrs": bp[:nb + 1].copy(), "prec": prec[:nb].copy(), "cond": cond[:nb].copy(), "x": x,
            "xt": xt, "xh": xh}


def generate():
    build_shim()
    lib = C.CDLL(SHIM)
    lib.ref_jacobi_types.restype = C.c_int64
    out = {}
    for vt, (name, _) in VT.items():
        rp, ci, vals, b = inputs(vt)
        out[f"{name}/row_ptrs"], out[f"{name}/col_idxs"], out[f"{name}/values"], out[f"{name}/b"] = rp, ci, vals, b
        for max_bs in SIZES[vt]:
            for tag, req, acc in REQUESTS:
                res = reference(lib, vt, rp, ci, vals, b, max_bs, req, acc)
                res["request"] = np.asarray(req, np.uint8)      # replicated over the blocks (jacobi.cpp:386-396)
                res["accuracy"] = np.asarray([acc], np.float64)
                for k, v in res.items():
                    out[f"{name}/{max_bs}/{tag}/{k}"] = v
    return out


def main():
    out = generate()
    if "--check" in sys.argv:
        old = np.load(OUT)
        assert sorted(old.files) == sorted(out), "different set of arrays"
        for k in old.files:
            assert old[k].tobytes() == out[k].tobytes(), k
        print(f"fixture == live reference ({len(old.files)} arrays)")
        return
    np.savez_compressed(OUT, **out)
    kinds = {}
    for k, v in out.items():
        if k.endswith("/prec"):
            kinds[k] = {f"{int(p):#04x}": int((v == p).sum()) for p in np.unique(v)}
    for k in sorted(kinds):
        print(k, kinds[k])
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
