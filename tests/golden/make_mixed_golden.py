#!/usr/bin/env python3
"""Generates tests/golden/mixed_spmv.npz with the UNMODIFIED reference built with
GINKGO_MIXED_PRECISION (oracle/_ref/mixed: oracle/build_ref_mixed.py + oracle/ref_shim_mixed.cpp):
Csr / Ell apply on gko::ReferenceExecutor for the six non-uniform (matrix, input, output) triples of
float32 / float64, c = A b and c = alpha A b + beta c, int32 and int64 indices, 2 right-hand sides.

    python tests/golden/make_mixed_golden.py          # write the fixture
    python tests/golden/make_mixed_golden.py --check  # oracle == live reference on fresh inputs

(A process of its own: this flavor's libginkgo.so must not meet the plain flavor's in one process.)
The fixture pins oracle/gko_oracle_mixed.inc (tests/test_mixed_cpu.py) and travels to the GPU box."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
SHIM = os.path.join(ROOT, "oracle", "_ref", "mixed", "libgko_ref_shim_mixed.so")

DT = {0: np.float64, 1: np.float32}
TRIPLES = [(m, i, o) for m in (0, 1) for i in (0, 1) for o in (0, 1) if not (m == i == o)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def inputs(seed, n_rows=37, n_cols=29):
    """rows of 0 ... 12 entries (an empty row included), values that are NOT exactly representable
    in float32, so that every narrowing and widening rounds"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 13, n_rows)
    lens[5] = 0
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ci = np.concatenate([np.sort(rng.choice(n_cols, l, replace=False)) for l in lens]).astype(np.int64)
    vals = rng.uniform(-1, 1, int(rp[-1])) * 10.0 ** rng.integers(-3, 4, int(rp[-1]))
    b = rng.uniform(-1, 1, (n_cols, 2))
    c = rng.uniform(-1, 1, (n_rows, 2))
    return rp, ci, vals, b, c


def to_ell(rp, ci, vals):
    n = len(rp) - 1
    k = int(np.diff(rp).max())
    stride = n + 3                      # a stride larger than the number of rows
    cols = np.full(k * stride, -1, ci.dtype)
    ev = np.zeros(k * stride, vals.dtype)
    for r in range(n):
        for j, p in enumerate(range(rp[r], rp[r + 1])):
            cols[r + j * stride] = ci[p]
            ev[r + j * stride] = vals[p]
    return k, stride, cols, ev


def live(lib, fmt, triple, idx64, n_cols, rp, ci, vals, b, c, alpha=None, beta=None):
    m, i, o = triple
    idt = np.int64 if idx64 else np.int32
    v = np.ascontiguousarray(vals, DT[m])
    bb = np.ascontiguousarray(b, DT[i])
    out = np.array(c, DT[o], order="C", copy=True)
    n_rows = out.shape[0]
    if fmt == 0:
        ptrs, cols, k, stride = rp.astype(idt), ci.astype(idt), 0, 0
    else:
        k, stride, cols, v = to_ell(rp, ci.astype(idt), v)
        ptrs = np.zeros(1, idt)
    a_ = None if alpha is None else np.array([alpha], DT[m])
    b_ = None if beta is None else np.array([beta], DT[o])
    rc = lib.ref_mixed_apply(fmt, m, i, o, int(idx64), C.c_int64(n_rows), C.c_int64(n_cols), C.c_int64(k),
                             C.c_int64(stride), _p(ptrs), _p(cols), _p(v), None if a_ is None else _p(a_),
                             _p(bb), C.c_int64(bb.shape[1]), None if b_ is None else _p(b_), _p(out))
    assert rc == 0
    return out


def oracle(fmt, triple, idx64, rp, ci, vals, b, c, alpha=None, beta=None):
    from oracle import gko_oracle as o
    m, i, ot = triple
    idt = np.int64 if idx64 else np.int32
    v = np.ascontiguousarray(vals, DT[m])
    bb = np.ascontiguousarray(b, DT[i])
    cc = np.array(c, DT[ot])
    if fmt == 0:
        return o.csr_spmv_mixed(rp.astype(idt), ci.astype(idt), v, bb, DT[ot], alpha, beta, cc)
    k, stride, cols, ev = to_ell(rp, ci.astype(idt), v)
    return o.ell_spmv_mixed(len(rp) - 1, k, stride, cols, ev, bb, DT[ot], alpha, beta, cc)


ALPHA, BETA = -1.3, 0.7


def main():
    if not os.path.exists(SHIM):
        print("oracle/_ref/mixed is not built (python oracle/build_ref_mixed.py)")
        return 1
    lib = C.CDLL(SHIM)
    if "--check" in sys.argv:
        bad = 0
        for seed in range(5):
            rp, ci, vals, b, c = inputs(100 + seed, 60 + seed, 45)
            for fmt in (0, 1):
                for t in TRIPLES:
                    for idx64 in (False, True):
                        for ab in ((None, None), (ALPHA, BETA), (0.5, 0.0)):
                            want = live(lib, fmt, t, idx64, 45, rp, ci, vals, b, c, *ab)
                            got = oracle(fmt, t, idx64, rp, ci, vals, b, c, *ab)
                            if got.dtype != want.dtype or not np.array_equal(got.view(np.uint8), want.view(np.uint8)):
                                bad += 1
                                print("MISMATCH", fmt, t, idx64, ab)
        print("oracle == live reference (GINKGO_MIXED_PRECISION)" if bad == 0 else f"{bad} mismatches")
        return 1 if bad else 0
    rp, ci, vals, b, c = inputs(2024)
    out = {"row_ptrs": rp, "col_idxs": ci, "vals": vals, "b": b, "c": c, "alpha": np.float64(ALPHA),
           "beta": np.float64(BETA)}
    for fmt, fn in ((0, "csr"), (1, "ell")):
        for t in TRIPLES:
            for idx64 in (False, True):
                key = f"{fn}_{t[0]}{t[1]}{t[2]}_{'i64' if idx64 else 'i32'}"
                out[key + "_spmv"] = live(lib, fmt, t, idx64, b.shape[0], rp, ci, vals, b, c)
                out[key + "_adv"] = live(lib, fmt, t, idx64, b.shape[0], rp, ci, vals, b, c, ALPHA, BETA)
    np.savez_compressed(os.path.join(HERE, "mixed_spmv.npz"), **out)
    print("wrote mixed_spmv.npz:", len(out), "arrays")
    return 0


if __name__ == "__main__":
    sys.exit(main())
