"""(HISTORICAL: the variants this script switched between were removed after the measurement,
profiles/r06/r06_waves_per_workgroup.txt.)  A/B of the row-segment CSR kernel's launch shapes (development tool): waves per workgroup (GKOC_TUNE_CSR_SHORT_ROWS
= 6 / 7 / 8: four / two / eight) and the XCD-contiguous order (GKOC_TUNE_CSR_XCD_MAP) on matrices with short and
long rows; one process, times by HIP events, results compared bit for bit with the default launch."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import scipy.sparse as sp
import torch

import ginkgo_amd as g
from ginkgo_amd import workloads as wl

ex = g.Cdna4Executor.create(0)
L = g._lib.lib()


def t(a, x, y, reps=20):
    for _ in range(5):
        a.apply(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        a.apply(x, y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def irregular():
    n = 4000000
    rp, ci, va = wl.irregular_rows(n)
    return g.Csr.from_scipy(ex, sp.csr_matrix((va, ci, rp), shape=(n, n)))


def flan():
    return wl.flan_like_csr(ex, 80) if hasattr(wl, "flan_like_csr") else None


cases = [("irregular 4M", irregular), ("5pt 4096^2", lambda: g.stencil_csr(ex, 2, 4096, restricted=True)),
         ("5pt 2048^2", lambda: g.stencil_csr(ex, 2, 2048, restricted=True)),
         ("7pt 200^3", lambda: g.stencil_csr(ex, 3, 200, restricted=True)),
         ("27pt 128^3", lambda: g.stencil_csr(ex, 3, 128)), ("27pt 256^3", lambda: g.stencil_csr(ex, 3, 256))]
only = sys.argv[1:] or None
for name, mk in cases:
    if only and not any(o in name for o in only):
        continue
    a = mk()
    n = a.size[0]
    x = g.Dense.from_numpy(ex, np.random.default_rng(1).uniform(-1, 1, n))
    y = g.Dense.create(ex, (n, 1))
    a.apply(x, y)
    ref = y.to_numpy().tobytes()
    row = []
    for wpb, xcd in ((0, 0), (0, 1), (7, 0), (6, 0), (8, 0), (7, 1), (6, 1), (8, 1)):
        L.gkoc_tune_set(C.c_int(14), C.c_int64(wpb))
        L.gkoc_tune_set(C.c_int(0), C.c_int64(xcd))
        us = t(a, x, y)
        same = y.to_numpy().tobytes() == ref
        row.append(f"wpb {dict([(0, 1), (7, 2), (6, 4), (8, 8)])[wpb]}{' xcd' if xcd else ''}: {us:.1f}{'' if same else ' BITS DIFFER'}")
    L.gkoc_tune_set(C.c_int(14), C.c_int64(0))
    L.gkoc_tune_set(C.c_int(0), C.c_int64(0))
    print(f"{name:14s} " + " | ".join(row), flush=True)
    del a, x, y
