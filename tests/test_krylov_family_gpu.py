"""GPU parity of the Bicgstab / Cgs / Fcg / PipeCg kernels and drivers (SURVEY 8(f)
rank 3), through the C ABI, against the oracle.

Mirrors reference/test/solver/{bicgstab,cgs,fcg,pipe_cg}_kernels.cpp (known answers,
restated in krylov_family_cases.py) and test/solver/{bicgstab,cgs,fcg,pipe_cg}_kernels.cpp
(reference vs device on seeded operands: several columns, strides, a stopped column,
zero denominators).  Bars: every step kernel bit-exact (vectors AND the scalars /
stopping status it writes); full solves: same iteration count +-1 (the dots are tree
sums), relative solution error <= 1e-8, and the golden solutions of the reference."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import krylov_family_cases as kc
from krylov_family_abi import KERNELS

pytestmark = pytest.mark.gpu

ALL = [(s, k) for s, ks in KERNELS.items() for k in ks]


def run_device(gexec, solver, kernel, arrays, strides=None):
    """arrays: name -> numpy (C order); runs gkoc_<solver>_<kernel> on device copies (vector
    operands with leading dimension strides[name], default cols) and writes the results back"""
    from ginkgo_amd._lib import VT, call
    import ginkgo_amd as g
    spec = KERNELS[solver][kernel]
    shape = next((arrays[n] for n, k in spec if k in "Vv"), None)
    if shape is None:                      # scalars only: (stream, cols, ...)
        rows, cols = None, len(next(arrays[n] for n, k in spec if k in "Ss"))
    else:
        rows, cols = shape.shape
    dev, args = {}, []
    for name, kind in spec:
        a = arrays[name]
        if kind in "Vv":
            ld = (strides or {}).get(name, cols)
            d = g.Dense.from_numpy(gexec, a, ld)
            dev[name] = d
            args += [d.values, d.ld]
        else:
            t = gexec.to_device(a)
            dev[name] = t
            args.append(t)
    vt = VT[next(d.dtype for (n, k), d in zip(spec, dev.values()) if k in "VvSs")]
    dims = [cols] if rows is None else [rows, cols]
    call(f"gkoc_{solver}_{kernel}_" + vt, gexec.stream, *dims, *args)
    gexec.synchronize()
    for name, kind in spec:
        if kind in "vstpU":
            d = dev[name]
            arrays[name][...] = d.to_numpy() if kind == "v" else d.cpu().numpy()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_known_answers(gexec, dtype):
    for solver, kernel, inp, stop, exp in kc.CASES:
        arr = kc.materialise(solver, kernel, inp, stop, dtype)
        run_device(gexec, solver, kernel, arr)
        kc.check(arr, exp)


@pytest.mark.parametrize("solver,kernel", ALL)
@pytest.mark.parametrize("rows,cols,dtype", [(597, 3, np.float64), (100003, 1, np.float64),
                                             (4096, 1, np.float32), (33, 5, np.float32)])
def test_step_kernels_bit_exact(gexec, oracle, solver, kernel, rows, cols, dtype):
    for variant, (zero_col, stopped_col) in enumerate(((None, None), (0, 1), (1, 0), (0, None))):
        ref = kc.random_case(solver, kernel, rows, cols, dtype, 100 * rows + variant, zero_col, stopped_col)
        if kernel == "finalize":
            ref["stop_status"][:] = [kc.STOPPED, kc.FINAL, kc.RUN][variant % 3]
        dev = {k: v.copy() for k, v in ref.items()}
        oracle.krylov_step(f"{solver}_{kernel}", rows, cols, *ref.values())
        run_device(gexec, solver, kernel, dev)
        for name, kind in KERNELS[solver][kernel]:
            assert np.array_equal(dev[name], ref[name]), (solver, kernel, name, variant)


@pytest.mark.parametrize("solver,kernel", ALL)
def test_step_kernels_strided(gexec, oracle, solver, kernel):
    """every operand with its own leading dimension (x and b are the caller's)"""
    rows, cols = 211, 4
    ref = kc.random_case(solver, kernel, rows, cols, np.float64, 7, None, 2)
    dev = {k: v.copy() for k, v in ref.items()}
    strides = {n: cols + 1 + i % 3 for i, (n, k) in enumerate(KERNELS[solver][kernel]) if k in "Vv"}
    oracle.krylov_step(f"{solver}_{kernel}", rows, cols, *ref.values())
    run_device(gexec, solver, kernel, dev, strides)
    for name, kind in KERNELS[solver][kernel]:
        assert np.array_equal(dev[name], ref[name]), (solver, kernel, name)


def _solver(g, kind):
    return {"bicgstab": g.Bicgstab, "cgs": g.Cgs, "fcg": g.Fcg, "pipe_cg": g.PipeCg}[kind]


def _solve(g, gexec, kind, a, rhs, max_iters, reduction, bs, x0=None, baseline=None):
    rn = g.stop.ResidualNorm.build().with_reduction_factor(reduction)
    if baseline is not None:
        rn = rn.with_baseline(baseline)
    f = _solver(g, kind).build().with_criteria(g.stop.Iteration.build().with_max_iters(max_iters), rn)
    if bs:
        f = f.with_preconditioner(g.Jacobi.build().with_max_block_size(bs))
    s = f.on(gexec).generate(a)
    x = g.Dense.from_numpy(gexec, np.zeros(len(rhs)) if x0 is None else x0)
    s.apply(g.Dense.from_numpy(gexec, rhs), x)
    return x.to_numpy()[:, 0], s


def test_solves_match_reference_golden(gexec):
    """tests/golden/krylov_family.npz holds the reference's own solutions"""
    import ginkgo_amd as g
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "krylov_family.npz"))
    for mname, kinds in (("sym", ("bicgstab", "cgs", "fcg", "pipe_cg")), ("nonsym", ("bicgstab", "cgs"))):
        rp, ci, v, rhs = (gold[f"{mname}_{k}"] for k in ("row_ptrs", "cols", "vals", "rhs"))
        n = len(rp) - 1
        a = g.Csr.from_arrays(gexec, (n, n), rp, ci, v)
        for kind in kinds:
            for bs in (0, 1, 8):
                x, s = _solve(g, gexec, kind, a, rhs, 400, 1e-9, bs)
                it_ref, rn_ref = gold[f"{mname}_{kind}_{bs}_it_rn"]
                xr = gold[f"{mname}_{kind}_{bs}_x"]
                assert s.has_converged and abs(s.num_iterations - int(it_ref)) <= 1, (mname, kind, bs)
                assert np.linalg.norm(x - xr) <= 1e-8 * np.linalg.norm(xr), (mname, kind, bs)
            # fixed iteration count from a non-zero guess: the iterates themselves agree
            x, s = _solve(g, gexec, kind, a, rhs, 6, 1e-30, 8, x0=np.full(n, 0.5),
                          baseline=g.stop.mode.initial_resnorm)
            xr = gold[f"{mname}_{kind}_lim_x"]
            assert s.num_iterations == 6 and not s.has_converged
            assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr), (mname, kind)
            assert abs(s.residual_norm - gold[f"{mname}_{kind}_lim_it_rn"][1]) <= 1e-9 * s.residual_norm


@pytest.mark.parametrize("kind", ["bicgstab", "cgs", "fcg", "pipe_cg"])
def test_solves_27pt_vs_oracle(gexec, oracle, kind):
    import ginkgo_amd as g
    grid = 24
    rp, ci, v = oracle.stencil_csr(3, grid)
    n = grid ** 3
    a = g.stencil_csr(gexec, 3, grid)
    rhs = np.ones(n)
    for bs, pre in ((0, None), (8, "block")):
        xo, ito, rno = oracle.krylov_solve(kind, rp, ci, v, rhs, max_iters=300, reduction=1e-10,
                                           precond=pre, max_block_size=max(bs, 1))
        x, s = _solve(g, gexec, kind, a, rhs, 300, 1e-10, bs)
        assert s.has_converged and abs(s.num_iterations - ito) <= 1, (kind, bs, s.num_iterations, ito)
        assert np.linalg.norm(x - xo) <= 1e-8 * np.linalg.norm(xo)
        # true residual of the device solution
        r = rhs - oracle.csr_spmv(rp, ci, v, x)
        assert np.linalg.norm(r) <= 2e-10 * np.linalg.norm(rhs)


def test_f32_and_multiple_rhs(gexec, oracle):
    """the drivers are value-type and column-count generic (kernels bit-exact per column)"""
    import ginkgo_amd as g
    rp, ci, v = oracle.stencil_csr(2, 40, True)
    n = 1600
    a = g.Csr.from_arrays(gexec, (n, n), rp, ci, v)
    rhs = np.random.default_rng(2).uniform(-1, 1, (n, 3))
    for kind in ("bicgstab", "cgs", "fcg", "pipe_cg"):
        f = _solver(g, kind).build().with_criteria(
            g.stop.Iteration.build().with_max_iters(500),
            g.stop.ResidualNorm.build().with_reduction_factor(1e-9))
        s = f.on(gexec).generate(a)
        x = g.Dense.from_numpy(gexec, np.zeros((n, 3)))
        s.apply(g.Dense.from_numpy(gexec, rhs), x)
        xs = x.to_numpy()
        for j in range(3):
            r = rhs[:, j] - oracle.csr_spmv(rp, ci, v, np.ascontiguousarray(xs[:, j]))
            assert np.linalg.norm(r) <= 1e-8 * np.linalg.norm(rhs[:, j]), (kind, j)
    a32 = g.Csr.from_arrays(gexec, (n, n), rp, ci, v.astype(np.float32))
    for kind in ("bicgstab", "fcg"):
        f = _solver(g, kind).build().with_criteria(
            g.stop.Iteration.build().with_max_iters(500),
            g.stop.ResidualNorm.build().with_reduction_factor(1e-4))
        s = f.on(gexec).generate(a32)
        x = g.Dense.from_numpy(gexec, np.zeros((n, 1), dtype=np.float32))
        s.apply(g.Dense.from_numpy(gexec, rhs[:, :1].astype(np.float32)), x)
        r = rhs[:, 0] - oracle.csr_spmv(rp, ci, v, x.to_numpy()[:, 0].astype(np.float64))
        assert s.has_converged and np.linalg.norm(r) <= 1e-3 * np.linalg.norm(rhs[:, 0])


# ------------------------------------------------------------ Ir / Chebyshev
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_chebyshev_kernels(gexec, oracle, dtype):
    """known answers of reference/test/solver/chebyshev_kernels.cpp, then seeded operands
    with strides against the oracle, bit-exact"""
    import ginkgo_amd as g
    from ginkgo_amd._lib import VT, call
    suf = "f64" if dtype == np.float64 else "f32"
    for kernel in ("init_update", "update"):
        inner, upd, out = (g.Dense.from_numpy(gexec, np.array(m, dtype=dtype))
                           for m in (kc.CHEB_INNER, kc.CHEB_UPDATE, kc.CHEB_OUTPUT))
        coeffs = [C.c_double(0.5)] + ([C.c_double(0.25)] if kernel == "update" else [])
        call(f"gkoc_chebyshev_{kernel}_{suf}", gexec.stream, 3, 3, *coeffs, inner.values, inner.ld,
             upd.values, upd.ld, out.values, out.ld)
        kc.check_chebyshev(kernel, inner.to_numpy(), upd.to_numpy(), out.to_numpy())
    rng = np.random.default_rng(3)
    for rows, cols, lds in ((100003, 1, (1, 1, 1)), (597, 3, (4, 3, 5))):
        for kernel in ("init_update", "update"):
            host = [rng.uniform(-1, 1, (rows, cols)).astype(dtype) for _ in range(3)]
            dev = [g.Dense.from_numpy(gexec, h, ld) for h, ld in zip(host, lds)]
            coeffs = [0.37] + ([-0.61] if kernel == "update" else [])
            f = getattr(oracle.lib(), f"oracle_chebyshev_{kernel}_{suf}")
            f(C.c_int64(rows), C.c_int64(cols), C.c_int64(cols), *map(C.c_double, coeffs),
              *[oracle._p(h) for h in host])
            call(f"gkoc_chebyshev_{kernel}_{suf}", gexec.stream, rows, cols, *map(C.c_double, coeffs),
                 *[t for d in dev for t in (d.values, d.ld)])
            for h, d in zip(host, dev):
                assert np.array_equal(d.to_numpy(), h), (kernel, rows)


def test_ir_chebyshev_match_reference_golden(gexec, oracle):
    import ginkgo_amd as g
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "stationary.npz"))
    rp, ci, v, rhs = (gold[k] for k in ("row_ptrs", "cols", "vals", "rhs"))
    n = len(rp) - 1
    a = g.Csr.from_arrays(gexec, (n, n), rp, ci, v)

    def solve(cls, bs, x0, max_iters, reduction, baseline=None, **params):
        rn = g.stop.ResidualNorm.build().with_reduction_factor(reduction)
        if baseline is not None:
            rn = rn.with_baseline(baseline)
        f = cls.build().with_criteria(g.stop.Iteration.build().with_max_iters(max_iters), rn)
        for k, val in params.items():
            f = getattr(f, "with_" + k)(val)
        if bs:
            f = f.with_preconditioner(g.Jacobi.build().with_max_block_size(bs))
        s = f.on(gexec).generate(a)
        x = g.Dense.from_numpy(gexec, x0)
        s.apply(g.Dense.from_numpy(gexec, rhs), x)
        return x.to_numpy()[:, 0], s

    for bs in (0, 1, 8):
        it_ref, rn_ref, relax = gold[f"ir_{bs}_it_rn"]
        x, s = solve(g.Ir, bs, np.zeros(n), 400, 1e-6, relaxation_factor=float(relax))
        assert s.has_converged and abs(s.num_iterations - int(it_ref)) <= 1, ("ir", bs, s.num_iterations, it_ref)
        assert np.linalg.norm(x - gold[f"ir_{bs}_x"]) <= 1e-9 * np.linalg.norm(x)
        it_ref, rn_ref, f0, f1 = gold[f"chebyshev_{bs}_it_rn"]
        x, s = solve(g.Chebyshev, bs, np.full(n, 0.1), 400, 1e-6, foci=(float(f0), float(f1)))
        assert s.has_converged and abs(s.num_iterations - int(it_ref)) <= 1, ("chebyshev", bs)
        assert np.linalg.norm(x - gold[f"chebyshev_{bs}_x"]) <= 1e-9 * np.linalg.norm(x)
    # fixed iteration count: no reduction order is involved in the iterates (SpMV,
    # Jacobi and the updates are all bit-exact) => identical bits
    for cls, kind, kw in ((g.Ir, "ir", dict(relaxation_factor=0.9)), (g.Chebyshev, "chebyshev", dict(foci=(0.02, 2.0)))):
        x, s = solve(cls, 8, np.full(n, 0.5), 7, 1e-30, baseline=g.stop.mode.initial_resnorm, **kw)
        assert s.num_iterations == 7 and not s.has_converged
        assert np.array_equal(x, gold[f"{kind}_lim_x"]), kind
        assert abs(s.residual_norm - gold[f"{kind}_lim_it_rn"][1]) <= 1e-12 * s.residual_norm


# -------------------------------------------------------------------- Bicg
def test_bicg_matches_reference_golden(gexec, oracle):
    """Bicg drives A^T (csr transpose) and M^T (Jacobi transpose) next to A and M"""
    import ginkgo_amd as g
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "bicg.npz"))
    rp, ci, v, rhs = (gold[k] for k in ("row_ptrs", "cols", "vals", "rhs"))
    n = len(rp) - 1
    a = g.Csr.from_arrays(gexec, (n, n), rp, ci, v)
    for bs in (0, 1, 8):
        x, s = _solve_kind(g, gexec, g.Bicg, a, rhs, 400, 1e-9, bs)
        it_ref, _ = gold[f"bicg_{bs}_it_rn"]
        assert s.has_converged and abs(s.num_iterations - int(it_ref)) <= 1, (bs, s.num_iterations, it_ref)
        assert np.linalg.norm(x - gold[f"bicg_{bs}_x"]) <= 1e-8 * np.linalg.norm(x)
    x, s = _solve_kind(g, gexec, g.Bicg, a, rhs, 6, 1e-30, 8, x0=np.full(n, 0.5),
                       baseline=g.stop.mode.initial_resnorm)
    assert s.num_iterations == 6
    assert np.linalg.norm(x - gold["bicg_lim_x"]) <= 1e-10 * np.linalg.norm(x)
    # the transposed block-Jacobi: M^T b against the oracle (transposed blocks), bit-exact,
    # also for reduced and adaptive storage
    nb, ptrs = oracle.jacobi_find_blocks(rp, ci, 8)
    scheme = oracle.jacobi_storage_scheme(8)
    full = oracle.jacobi_generate(rp, ci, v, nb, scheme, ptrs)
    bo, go, gp = scheme
    stride = bo << gp
    tr = full.copy()
    for blk in range(nb):
        base = go * (blk >> gp) + bo * (blk & ((1 << gp) - 1))
        bsz = int(ptrs[blk + 1] - ptrs[blk])
        for r in range(bsz):
            for c in range(bsz):
                tr[base + r + c * stride] = full[base + c + r * stride]
    b = np.random.default_rng(2).uniform(-1, 1, n)
    m = g.Jacobi.build().with_max_block_size(8).on(gexec).generate(a)
    y = g.Dense.create(gexec, (n, 1))
    m.transpose().apply(g.Dense.from_numpy(gexec, b), y)
    assert np.array_equal(y.to_numpy()[:, 0], oracle.jacobi_apply(nb, scheme, ptrs, tr, b))
    for prec in ((0, 1), (0, 2), (2, 0)):
        code = (prec[0] << 4) | prec[1]
        mp = g.Jacobi.build().with_max_block_size(8).with_storage_optimization(*prec).on(gexec).generate(a)
        mp.transpose().apply(g.Dense.from_numpy(gexec, b), y)
        want = oracle.jacobi_apply_stored(nb, scheme, ptrs, oracle.jacobi_convert_storage(nb, scheme, tr, code), code, b)
        assert np.array_equal(y.to_numpy()[:, 0], want), prec


def _solve_kind(g, gexec, cls, a, rhs, max_iters, reduction, bs, x0=None, baseline=None):
    rn = g.stop.ResidualNorm.build().with_reduction_factor(reduction)
    if baseline is not None:
        rn = rn.with_baseline(baseline)
    f = cls.build().with_criteria(g.stop.Iteration.build().with_max_iters(max_iters), rn)
    if bs:
        f = f.with_preconditioner(g.Jacobi.build().with_max_block_size(bs))
    s = f.on(gexec).generate(a)
    x = g.Dense.from_numpy(gexec, np.zeros(len(rhs)) if x0 is None else x0)
    s.apply(g.Dense.from_numpy(gexec, rhs), x)
    return x.to_numpy()[:, 0], s


def test_gcr_matches_reference_golden(gexec, oracle):
    import ginkgo_amd as g
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "bicg.npz"))
    rp, ci, v, rhs = (gold[k] for k in ("row_ptrs", "cols", "vals", "rhs"))
    n = len(rp) - 1
    a = g.Csr.from_arrays(gexec, (n, n), rp, ci, v)
    for kd in (100, 6):
        for bs in (0, 8):
            f = g.Gcr.build().with_krylov_dim(kd).with_criteria(
                g.stop.Iteration.build().with_max_iters(400),
                g.stop.ResidualNorm.build().with_reduction_factor(1e-9))
            if bs:
                f = f.with_preconditioner(g.Jacobi.build().with_max_block_size(bs))
            s = f.on(gexec).generate(a)
            x = g.Dense.from_numpy(gexec, np.zeros(n))
            s.apply(g.Dense.from_numpy(gexec, rhs), x)
            it_ref, _ = gold[f"gcr_{kd}_{bs}_it_rn"]
            assert s.has_converged and abs(s.num_iterations - int(it_ref)) <= 1, (kd, bs, s.num_iterations, it_ref)
            xr = gold[f"gcr_{kd}_{bs}_x"]
            assert np.linalg.norm(x.to_numpy()[:, 0] - xr) <= 1e-7 * np.linalg.norm(xr)
            r = rhs - oracle.csr_spmv(rp, ci, v, x.to_numpy()[:, 0])
            assert np.linalg.norm(r) <= 2e-9 * np.linalg.norm(rhs)


def test_minres_matches_reference_golden(gexec, oracle):
    """symmetric indefinite operator; ResidualNorm recomputes b - A x (no residual handed over),
    ImplicitResidualNorm uses tau"""
    import ginkgo_amd as g
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "bicg.npz"))
    rp, ci, v, rhs = (gold[k] for k in ("m_row_ptrs", "m_cols", "m_vals", "m_rhs"))
    n = len(rp) - 1
    a = g.Csr.from_arrays(gexec, (n, n), rp, ci, v)
    for bs in (0, 1):
        x, s = _solve_kind(g, gexec, g.Minres, a, rhs, 400, 1e-9, bs)
        it_ref, _ = gold[f"minres_{bs}_it_rn"]
        assert s.has_converged and abs(s.num_iterations - int(it_ref)) <= 1, (bs, s.num_iterations, it_ref)
        xr = gold[f"minres_{bs}_x"]
        assert np.linalg.norm(x - xr) <= 1e-7 * np.linalg.norm(xr)
    x, s = _solve_kind(g, gexec, g.Minres, a, rhs, 7, 1e-30, 0, x0=np.full(n, 0.5),
                       baseline=g.stop.mode.initial_resnorm)
    assert s.num_iterations == 7
    assert np.linalg.norm(x - gold["minres_lim_x"]) <= 1e-10 * np.linalg.norm(x)
    f = g.Minres.build().with_criteria(g.stop.Iteration.build().with_max_iters(400),
                                       g.stop.ImplicitResidualNorm.build().with_reduction_factor(1e-9))
    s = f.on(gexec).generate(a)
    x = g.Dense.from_numpy(gexec, np.zeros(n))
    s.apply(g.Dense.from_numpy(gexec, rhs), x)
    r = rhs - oracle.csr_spmv(rp, ci, v, x.to_numpy()[:, 0])
    assert s.has_converged and np.linalg.norm(r) <= 1e-6 * np.linalg.norm(rhs)
