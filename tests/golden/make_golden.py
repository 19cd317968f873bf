#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the UNMODIFIED reference (oracle/_ref, i.e.
Ginkgo 1.12.0's ReferenceExecutor built from /root/reference by
oracle/build_ref.py + oracle/build_shim.py).  Run in the build container:

    python tests/golden/make_golden.py

The fixtures pin oracle/gko_oracle.c (tests/test_oracle_cpu.py) and travel to
the GPU box, where /root/reference does not exist.  Inputs are seeded; the
script is deterministic."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_shim as ref  # noqa: E402
from util import random_csr  # noqa: E402


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name), **arrays)
    print("wrote", name, {k: getattr(v, "shape", v) for k, v in arrays.items()})


def nonsymmetric_5pt(o, grid):
    """5-pt stencil with the upper couplings scaled by 0.6 and the lower by 1.3
    (an upwinded convection-diffusion pattern): non-symmetric, diagonally dominant"""
    rp, ci, v = o.stencil_csr(2, grid, True)
    v = v.copy()
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    v[ci > rows] *= 0.6
    v[ci < rows] *= 1.3
    return rp, ci, v


def krylov_family():
    """Bicgstab / Cgs / Fcg / PipeCg of the reference"""
    from oracle import gko_oracle as o
    rng = np.random.default_rng(2024)
    arrays = {}
    mats = {"sym": o.stencil_csr(3, 10), "nonsym": nonsymmetric_5pt(o, 32)}
    for mname, (rp, ci, v) in mats.items():
        n = len(rp) - 1
        h = ref.CsrHandle("reference", rp, ci, v)
        rhs = rng.uniform(-1, 1, n)
        arrays[f"{mname}_row_ptrs"], arrays[f"{mname}_cols"] = rp, ci
        arrays[f"{mname}_vals"], arrays[f"{mname}_rhs"] = v, rhs
        kinds = ("bicgstab", "cgs", "fcg", "pipe_cg") if mname == "sym" else ("bicgstab", "cgs")
        for kind in kinds:
            for bs in (0, 1, 8):
                x, it, rn = h.krylov_solve(kind, rhs, max_iters=400, reduction=1e-9,
                                           precond_block_size=bs)
                arrays[f"{mname}_{kind}_{bs}_x"] = x
                arrays[f"{mname}_{kind}_{bs}_it_rn"] = np.array([it, rn])
            x, it, rn = h.krylov_solve(kind, rhs, x0=np.full(n, 0.5), max_iters=6, reduction=1e-30,
                                       baseline="initial_resnorm", precond_block_size=8)
            arrays[f"{mname}_{kind}_lim_x"] = x
            arrays[f"{mname}_{kind}_lim_it_rn"] = np.array([it, rn])
    save("krylov_family.npz", **arrays)


def coo_hybrid():
    """Coo apply / apply2 and Csr -> Hybrid of the reference"""
    rng = np.random.default_rng(77)
    rp, ci, v = random_csr(532, 231, 0.03, seed=5)       # the size of test/matrix/csr_kernels2.cpp
    rows = np.repeat(np.arange(532, dtype=np.int32), np.diff(rp))
    b = rng.uniform(-1, 1, (231, 3))
    c0 = rng.uniform(-1, 1, (532, 3))
    perm = rng.permutation(len(v))
    arrays = dict(row_ptrs=rp, rows=rows, cols=ci, vals=v, b=b, c0=c0, perm=perm)
    for mode in ("spmv", "advanced_spmv", "spmv2", "advanced_spmv2"):
        arrays[mode] = ref.coo_apply(mode, 532, 231, rows, ci, v, b, 2.0, -1.0, c0)
        arrays[mode + "_shuffled"] = ref.coo_apply(mode, 532, 231, rows[perm], ci[perm], v[perm], b,
                                                   2.0, -1.0, c0)
    h = ref.CsrHandle("reference", rp, ci, v, n_cols=231)
    for lim in (0, 4, 9, 1000):
        k, st, ec, ev, cr, cc, cv = h.to_hybrid(lim)
        arrays[f"hyb{lim}_shape"] = np.array([k, st])
        arrays[f"hyb{lim}_ell_cols"], arrays[f"hyb{lim}_ell_vals"] = ec, ev
        arrays[f"hyb{lim}_coo_rows"], arrays[f"hyb{lim}_coo_cols"] = cr, cc
        arrays[f"hyb{lim}_coo_vals"] = cv
        arrays[f"hyb{lim}_apply"] = h.hybrid_spmv(b)
    save("coo_hybrid.npz", **arrays)


def assembly():
    """device_matrix_data::{sort_row_major, remove_zeros, sum_duplicates} and
    Csr::read(device_matrix_data) of the reference"""
    rng = np.random.default_rng(2024)
    nnz, n_rows, n_cols = 6000, 532, 231
    rows = rng.integers(0, n_rows, nnz).astype(np.int32)
    cols = rng.integers(0, 40, nnz).astype(np.int32)          # few columns: many duplicates
    vals = rng.uniform(-1, 1, nnz)
    vals[rng.random(nnz) < 0.15] = 0.0
    vals[rng.random(nnz) < 0.05] = -0.0
    arrays = dict(shape=np.array([n_rows, n_cols]), rows=rows, cols=cols, vals=vals)
    for op in ("sort_row_major", "remove_zeros", "sum_duplicates", "csr"):
        r, c, v = ref.md_assemble(op, n_rows, n_cols, rows, cols, vals)
        arrays[op + "_rows"], arrays[op + "_cols"], arrays[op + "_vals"] = r, c, v
    save("assembly.npz", **arrays)


def stationary():
    """Ir and Chebyshev of the reference"""
    from oracle import gko_oracle as o
    rng = np.random.default_rng(31)
    rp, ci, v = o.stencil_csr(3, 10)
    n = len(rp) - 1
    h = ref.CsrHandle("reference", rp, ci, v)
    rhs = rng.uniform(-1, 1, n)
    arrays = dict(row_ptrs=rp, cols=ci, vals=v, rhs=rhs)
    # spectrum of the 27-pt operator: (0, 52); Jacobi-preconditioned: (0, 2)
    for bs, relax, foci in ((0, 0.035, (0.5, 52.0)), (1, 0.9, (0.02, 2.0)), (8, 0.9, (0.02, 2.0))):
        x, it, rn = h.stationary_solve("ir", rhs, max_iters=400, reduction=1e-6, precond_block_size=bs,
                                       relaxation=relax)
        arrays[f"ir_{bs}_x"], arrays[f"ir_{bs}_it_rn"] = x, np.array([it, rn, relax])
        x, it, rn = h.stationary_solve("chebyshev", rhs, x0=np.full(n, 0.1), max_iters=400, reduction=1e-6,
                                       precond_block_size=bs, foci=foci)
        arrays[f"chebyshev_{bs}_x"] = x
        arrays[f"chebyshev_{bs}_it_rn"] = np.array([it, rn, *foci])
    for kind, kw in (("ir", dict(relaxation=0.9)), ("chebyshev", dict(foci=(0.02, 2.0)))):
        x, it, rn = h.stationary_solve(kind, rhs, x0=np.full(n, 0.5), max_iters=7, reduction=1e-30,
                                       baseline="initial_resnorm", precond_block_size=8, **kw)
        arrays[f"{kind}_lim_x"], arrays[f"{kind}_lim_it_rn"] = x, np.array([it, rn])
    save("stationary.npz", **arrays)


def jacobi_storage():
    """block-Jacobi with a fixed storage_optimization: the reference's apply results"""
    from oracle import gko_oracle as o
    rng = np.random.default_rng(12)
    rp, ci, v = o.stencil_csr(3, 9)
    v = v * rng.uniform(0.5, 2.0, len(v))        # blocks of varied magnitude
    n = len(rp) - 1
    h = ref.CsrHandle("reference", rp, ci, v)
    b, x0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    arrays = dict(row_ptrs=rp, cols=ci, vals=v, b=b, x0=x0)
    for bs in (4, 8, 16):
        for p_, n_ in ((0, 1), (0, 2), (1, 0), (1, 1), (2, 0)):
            nb, scheme, ptrs, blocks = h.jacobi_generate_prec(bs, p_, n_)
            key = f"bs{bs}_p{p_}n{n_}"
            arrays[key + "_apply"] = h.jacobi_apply(b)
            arrays[key + "_apply_adv"] = h.jacobi_apply(b, 0.7, -1.1, x0)
    save("jacobi_storage.npz", **arrays)


def bicg():
    """Bicg of the reference (needs A^T and M^T) on a non-symmetric operator"""
    from oracle import gko_oracle as o
    rng = np.random.default_rng(99)
    rp, ci, v = nonsymmetric_5pt(o, 32)
    n = len(rp) - 1
    h = ref.CsrHandle("reference", rp, ci, v)
    rhs = rng.uniform(-1, 1, n)
    arrays = dict(row_ptrs=rp, cols=ci, vals=v, rhs=rhs)
    for bs in (0, 1, 8):
        x, it, rn = h.krylov_solve("bicg", rhs, max_iters=400, reduction=1e-9, precond_block_size=bs)
        arrays[f"bicg_{bs}_x"], arrays[f"bicg_{bs}_it_rn"] = x, np.array([it, rn])
    x, it, rn = h.krylov_solve("bicg", rhs, x0=np.full(n, 0.5), max_iters=6, reduction=1e-30,
                               baseline="initial_resnorm", precond_block_size=8)
    arrays["bicg_lim_x"], arrays["bicg_lim_it_rn"] = x, np.array([it, rn])
    trp, tc, tv = h.transpose()
    arrays["t_row_ptrs"], arrays["t_cols"], arrays["t_vals"] = trp, tc, tv
    for kd in (100, 6):
        for bs in (0, 8):
            x, it, rn = h.gcr_solve(rhs, krylov_dim=kd, max_iters=400, reduction=1e-9, precond_block_size=bs)
            arrays[f"gcr_{kd}_{bs}_x"], arrays[f"gcr_{kd}_{bs}_it_rn"] = x, np.array([it, rn])
    # Minres on a symmetric indefinite operator (27-pt Laplacian with a shifted diagonal)
    rp2, ci2, v2 = o.stencil_csr(3, 8)
    rows2 = np.repeat(np.arange(len(rp2) - 1), np.diff(rp2))
    v2 = v2.copy()
    v2[rows2 == ci2] -= 20.0
    h2 = ref.CsrHandle("reference", rp2, ci2, v2)
    rhs2 = rng.uniform(-1, 1, len(rp2) - 1)
    arrays.update(m_row_ptrs=rp2, m_cols=ci2, m_vals=v2, m_rhs=rhs2)
    for bs in (0, 1):
        x, it, rn = h2.krylov_solve("minres", rhs2, max_iters=400, reduction=1e-9, precond_block_size=bs)
        arrays[f"minres_{bs}_x"], arrays[f"minres_{bs}_it_rn"] = x, np.array([it, rn])
    x, it, rn = h2.krylov_solve("minres", rhs2, x0=np.full(len(rhs2), 0.5), max_iters=7, reduction=1e-30,
                                baseline="initial_resnorm")
    arrays["minres_lim_x"], arrays["minres_lim_it_rn"] = x, np.array([it, rn])
    save("bicg.npz", **arrays)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "bicg":
        return bicg()
    if len(sys.argv) > 1 and sys.argv[1] == "jacobi_storage":
        return jacobi_storage()
    if len(sys.argv) > 1 and sys.argv[1] == "stationary":
        return stationary()
    if len(sys.argv) > 1 and sys.argv[1] == "assembly":
        return assembly()
    if len(sys.argv) > 1 and sys.argv[1] == "krylov_family":
        return krylov_family()
    if len(sys.argv) > 1 and sys.argv[1] == "coo_hybrid":
        return coo_hybrid()
    assert ref.available(), "build oracle/_ref first (python oracle/build_ref.py; build_shim.py)"
    print("reference version", ref.version())
    rng = np.random.default_rng(2024)

    # --- CSR / ELL / SELL-P on the 532 x 231 matrix of test/matrix/csr_kernels2.cpp
    rp, ci, v = random_csr(532, 231, 0.08, 42, empty_rows=(0, 17, 531))
    b = rng.uniform(-1, 1, (231, 3))
    c0 = rng.uniform(-1, 1, (532, 3))
    h = ref.CsrHandle("reference", rp, ci, v, n_cols=231)
    k, stride, ecols, evals = h.to_ell()
    sets, lens, scols, svals = h.to_sellp(64, 1)
    save("spmv_532x231.npz", row_ptrs=rp, cols=ci, vals=v, b=b, c0=c0,
         y=h.spmv(b), y_adv=h.spmv(b, alpha=2.0, beta=-1.0, c=c0),
         ell_k=k, ell_stride=stride, ell_cols=ecols, ell_vals=evals,
         y_ell=h.ell_spmv(b), slice_sets=sets, slice_lengths=lens,
         sellp_cols=scols, sellp_vals=svals, y_sellp=h.sellp_spmv(b))

    # --- the reference's own stencil generator (single domain + subdomains)
    out = {}
    for name, nd, dims, pos, tls, restricted in [
            ("s3d_27pt_6", 3, [1, 1, 1], [0, 0, 0], 216, False),
            ("s3d_7pt_5", 3, [1, 1, 1], [0, 0, 0], 125, True),
            ("s2d_5pt_7", 2, [1, 1], [0, 0], 49, True),
            ("s2d_9pt_6", 2, [1, 1], [0, 0], 36, False),
            ("s3d_27pt_sub_2of3", 3, [3, 1, 1], [1, 0, 0], 72, False),
            ("s3d_27pt_sub_221", 3, [2, 2, 1], [1, 0, 0], 54, False),
            ("s2d_5pt_sub_1of2", 2, [2, 1], [1, 0], 32, True)]:
        r, c, vv, ls = ref.stencil_subdomain(nd, dims, pos, tls, restricted)
        out[name + "_rows"], out[name + "_cols"], out[name + "_vals"] = r, c, vv
        out[name + "_meta"] = np.array([nd, *dims, *([1] * (3 - len(dims))),
                                        *pos, *([0] * (3 - len(pos))), tls,
                                        int(restricted), ls], dtype=np.int64)
    save("stencil.npz", **out)

    # --- block-Jacobi + CG on the 27-pt 10^3 Laplacian and a block matrix
    from oracle import gko_oracle as o
    rp, ci, v = o.stencil_csr(3, 10)
    h = ref.CsrHandle("reference", rp, ci, v)
    rhs = rng.uniform(-1, 1, 1000)
    arrays = dict(row_ptrs=rp, cols=ci, vals=v, rhs=rhs)
    for bs in (1, 4, 8, 13, 32):
        if bs == 1:
            x, it, rn = h.cg_solve(np.ones(1000), precond_block_size=1)
        else:
            nb, scheme, ptrs, blocks = h.jacobi_generate(bs)
            arrays[f"jac{bs}_scheme"] = np.array(scheme)
            arrays[f"jac{bs}_ptrs"] = ptrs
            arrays[f"jac{bs}_blocks"] = blocks
            arrays[f"jac{bs}_apply"] = h.jacobi_apply(rhs)
            arrays[f"jac{bs}_apply_adv"] = h.jacobi_apply(rhs, 2.0, -1.0, np.ones(1000))
            x, it, rn = h.cg_solve(np.ones(1000), precond_block_size=bs)
        arrays[f"cg{bs}_x"] = x
        arrays[f"cg{bs}_iters"] = np.array([it])
        arrays[f"cg{bs}_resnorm"] = np.array([rn])
    x, it, rn = h.cg_solve(np.ones(1000), precond_block_size=0)
    arrays["cg0_x"], arrays["cg0_iters"], arrays["cg0_resnorm"] = x, np.array([it]), np.array([rn])
    x, it, rn = h.cg_solve(rhs, x0=np.full(1000, 0.5), max_iters=9, reduction=1e-30,
                           baseline="initial_resnorm", precond_block_size=8)
    arrays["cg_lim_x"], arrays["cg_lim_iters"] = x, np.array([it])
    for ortho in ("mgs", "cgs", "cgs2"):
        for kd, bs in ((100, 0), (7, 8)):
            xg, itg, rng_ = h.gmres_solve(rhs, krylov_dim=kd, ortho=ortho, max_iters=300,
                                          reduction=1e-9, precond_block_size=bs)
            arrays[f"gmres_{ortho}_{kd}_{bs}_x"] = xg
            arrays[f"gmres_{ortho}_{kd}_{bs}_iters"] = np.array([itg])
            arrays[f"gmres_{ortho}_{kd}_{bs}_resnorm"] = np.array([rng_])
    save("krylov_27pt_10.npz", **arrays)

    # --- block-Jacobi on a matrix with natural blocks of mixed size + pivoting
    import scipy.sparse as sp
    sizes = rng.integers(1, 7, 40)
    blocks = []
    for s in sizes:
        m = rng.uniform(-1, 1, (s, s))
        m[np.arange(s), np.arange(s)] = 0.01      # force row pivoting
        blocks.append(m + np.eye(s)[::-1] * 3)
    a = sp.block_diag(blocks).tocsr()
    a.sort_indices()
    rp, ci, v = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    h = ref.CsrHandle("reference", rp, ci, v)
    arrays = dict(row_ptrs=rp, cols=ci, vals=v)
    bvec = rng.uniform(-1, 1, (a.shape[0], 2))
    arrays["b"] = bvec
    for bs in (2, 6, 16):
        nb, scheme, ptrs, blks = h.jacobi_generate(bs)
        arrays[f"jac{bs}_scheme"] = np.array(scheme)
        arrays[f"jac{bs}_ptrs"] = ptrs
        arrays[f"jac{bs}_blocks"] = blks
        arrays[f"jac{bs}_apply"] = h.jacobi_apply(bvec)
    save("jacobi_blocks.npz", **arrays)

    # --- dense reductions (sequential reference order)
    x = rng.uniform(-1, 1, (5000, 3))
    y = rng.uniform(-1, 1, (5000, 3))
    save("dense.npz", x=x, y=y, dot=ref.dense_dot(x, y), norm2=ref.dense_norm2(x))


if __name__ == "__main__":
    main()
    if len(sys.argv) == 1:
        krylov_family()
        coo_hybrid()
        stationary()
        jacobi_storage()
        bicg()
        assembly()
