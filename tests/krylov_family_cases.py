"""Known-answer cases of the Bicgstab / Cgs / Fcg / PipeCg step kernels, restated from
the reference's own unit tests (2 x 2 operands filled with one value, scalar rows of
two values, column 1 stopped where the test stops it):
  reference/test/solver/bicg_kernels.cpp (KernelStep1 / Step1DivByZero / Step2 / Step2DivByZero)
  reference/test/solver/bicgstab_kernels.cpp:174-368
  reference/test/solver/cgs_kernels.cpp:170-279
  reference/test/solver/fcg_kernels.cpp:151-225
  reference/test/solver/pipe_cg_kernels.cpp:167-330
Each case: (solver, kernel, inputs, stop, expected).  inputs / expected map an
argument name to a fill value (operands) or a 2-list (scalar rows)."""
import numpy as np

from krylov_family_abi import KERNELS

STOPPED, FINAL, RUN = 0x01, 0x41, 0x00      # stopping_status: stop(1,false), stop(1,true), reset

CASES = [
    ("bicgstab", "step_1", dict(p=3, r=-2, v=1, rho=[2, 3], prev_rho=[7, 6], alpha=[7, 5], omega=[8, 3]),
     [RUN, STOPPED], dict(p=[[-3.25, 3.0], [-3.25, 3.0]])),
    ("bicgstab", "step_1", dict(p=3, r=-2, v=1, rho=[2, 2], prev_rho=[0, 0], alpha=[1, 1], omega=[1, 1]),
     [RUN, RUN], dict(p=-2.0)),
    ("bicgstab", "step_1", dict(p=3, r=-2, v=1, rho=[2, 2], prev_rho=[1, 1], alpha=[1, 1], omega=[0, 0]),
     [RUN, RUN], dict(p=-2.0)),
    ("bicgstab", "step_2", dict(s=5, r=-2, v=1, alpha=[0, 0], rho=[2, 3], beta=[8, 3]),
     [RUN, STOPPED], dict(s=[[-2.25, 5.0], [-2.25, 5.0]], alpha=[0.25, 0.0])),
    ("bicgstab", "step_2", dict(s=5, r=-2, v=1, alpha=[4, 4], rho=[1, 1], beta=[0, 0]),
     [RUN, RUN], dict(s=-2.0, alpha=[0.0, 0.0])),
    ("bicgstab", "step_3", dict(x=5, r=-2, s=1, y=4, z=-6, t=7, omega=[10, 10], beta=[2, 3], gamma=[8, 3],
                                alpha=[1, -2]),
     [RUN, STOPPED], dict(x=[[-15.0, 5.0], [-15.0, 5.0]], r=[[-27.0, -2.0], [-27.0, -2.0]], omega=[4.0, 10.0])),
    ("bicgstab", "step_3", dict(x=5, r=-2, s=1, y=4, z=-6, t=7, omega=[10, 10], beta=[0, 0], gamma=[8, 3],
                                alpha=[1, -2]),
     [RUN, RUN], dict(x=[[9.0, -3.0], [9.0, -3.0]], omega=[0.0, 0.0])),
    ("bicgstab", "finalize", dict(x=5, y=4, alpha=[1, -2]), [STOPPED, FINAL],
     dict(x=[[9.0, 5.0], [9.0, 5.0]], stop_status=[FINAL, FINAL])),
    ("bicg", "step_1", dict(p=3, z=-2, p2=3, z2=-2, rho=[2, 3], prev_rho=[8, 3]),
     [RUN, STOPPED], dict(p=[[-1.25, 3.0], [-1.25, 3.0]], p2=[[-1.25, 3.0], [-1.25, 3.0]])),
    ("bicg", "step_1", dict(p=3, z=-2, p2=3, z2=-2, rho=[1, 1], prev_rho=[0, 0]),
     [RUN, RUN], dict(p=-2.0, p2=-2.0)),
    ("bicg", "step_2", dict(x=-2, p=3, r=4, q=-5, r2=4, q2=-5, rho=[2, 3], beta=[8, 3]),
     [RUN, STOPPED], dict(x=[[-1.25, -2.0], [-1.25, -2.0]], r=[[5.25, 4.0], [5.25, 4.0]],
                          r2=[[5.25, 4.0], [5.25, 4.0]])),
    ("bicg", "step_2", dict(x=-2, p=3, r=4, q=-5, r2=4, q2=-5, rho=[1, 1], beta=[0, 0]),
     [RUN, RUN], dict(x=-2.0, r=4.0, r2=4.0)),
    ("gcr", "step_1", dict(x=1, residual=2, p=3, Ap=4, Ap_norm=[8, 0], rAp=[2, 5]),
     [RUN, RUN], dict(x=[[1.75, 1.0], [1.75, 1.0]], residual=[[1.0, 2.0], [1.0, 2.0]])),
    ("cgs", "step_1", dict(r=1, p=-2, q=3, u=-4, beta=[2, 2], rho_prev=[2, 3], rho=[-4, 4]),
     [RUN, STOPPED], dict(u=[[-5.0, -4.0], [-5.0, -4.0]], p=[[-19.0, -2.0], [-19.0, -2.0]], beta=[-2.0, 2.0])),
    ("cgs", "step_1", dict(r=1, p=-2, q=3, u=-4, beta=[2, 2], rho_prev=[0, 0], rho=[3, 3]),
     [RUN, RUN], dict(u=7.0, p=5.0, beta=[2.0, 2.0])),
    ("cgs", "step_2", dict(q=1, u=-2, v_hat=3, t=-4, alpha=[2, 2], gamma=[2, 3], rho=[-4, 4]),
     [RUN, STOPPED], dict(q=[[4.0, 1.0], [4.0, 1.0]], t=[[2.0, -4.0], [2.0, -4.0]], alpha=[-2.0, 2.0])),
    ("cgs", "step_2", dict(q=1, u=-2, v_hat=3, t=-4, alpha=[2, 2], gamma=[0, 0], rho=[-3, -3]),
     [RUN, RUN], dict(q=-8.0, t=-10.0, alpha=[2.0, 2.0])),
    ("cgs", "step_3", dict(r=1, t=-2, x=3, u_hat=-4, alpha=[2, 3]),
     [RUN, STOPPED], dict(r=[[5.0, 1.0], [5.0, 1.0]], x=[[-5.0, 3.0], [-5.0, 3.0]])),
    ("fcg", "step_1", dict(p=3, z=-2, rho_t=[2, 3], prev_rho=[8, 3]),
     [RUN, STOPPED], dict(p=[[-1.25, 3.0], [-1.25, 3.0]])),
    ("fcg", "step_1", dict(p=3, z=-2, rho_t=[1, 1], prev_rho=[0, 0]), [RUN, RUN], dict(p=-2.0)),
    ("fcg", "step_2", dict(x=-2, p=3, r=4, q=-5, t=8, rho=[2, 3], beta=[8, 3]),
     [RUN, STOPPED], dict(x=[[-1.25, -2.0], [-1.25, -2.0]], r=[[5.25, 4.0], [5.25, 4.0]],
                          t=[[1.25, 8.0], [1.25, 8.0]])),
    ("fcg", "step_2", dict(x=-2, p=3, r=4, q=-5, t=8, rho=[1, 1], beta=[0, 0]),
     [RUN, RUN], dict(x=-2.0, r=4.0, t=8.0)),
    ("pipe_cg", "step_1", dict(x=1, r=2, z1=3, z2=3, w=4, p=4, q=3, f=2, g=1, rho=[2, 3], beta=[8, 3]),
     [RUN, STOPPED], dict(x=[[2.0, 1.0], [2.0, 1.0]], r=[[1.25, 2.0], [1.25, 2.0]],
                          z1=[[2.5, 3.0], [2.5, 3.0]], z2=[[2.5, 3.0], [2.5, 3.0]],
                          w=[[3.75, 4.0], [3.75, 4.0]])),
    ("pipe_cg", "step_1", dict(x=1, r=2, z1=3, z2=3, w=4, p=4, q=3, f=2, g=1, rho=[1, 1], beta=[0, 0]),
     [RUN, RUN], dict(x=1.0, r=2.0, z1=3.0, w=4.0)),
    ("pipe_cg", "step_2", dict(z=1, w=2, m=3, n=4, p=4, q=3, f=2, g=1, rho=[-2, 3], prev_rho=[4, 3],
                               beta=[2, 3], delta=[5, 6]),
     [RUN, STOPPED], dict(beta=[4.5, 3.0], p=[[-1.0, 4.0], [-1.0, 4.0]], q=[[0.5, 3.0], [0.5, 3.0]],
                          f=[[2.0, 2.0], [2.0, 2.0]], g=[[3.5, 1.0], [3.5, 1.0]])),
    ("pipe_cg", "step_2", dict(z=1, w=2, m=3, n=4, p=4, q=3, f=2, g=1, rho=[-2, 3], prev_rho=[0, 0],
                               beta=[2, 3], delta=[5, 6]),
     [RUN, RUN], dict(beta=[5.0, 6.0], p=1.0, q=2.0, f=3.0, g=4.0)),
    ("pipe_cg", "step_2", dict(z=1, w=1, m=1, n=1, p=1, q=1, f=1, g=1, rho=[3, 3], prev_rho=[3, 6],
                               beta=[2, 4], delta=[2, 1]),
     [RUN, RUN], dict(beta=[2.0, 1.0], p=[[2.0, 1.5], [2.0, 1.5]], q=[[2.0, 1.5], [2.0, 1.5]],
                      f=[[2.0, 1.5], [2.0, 1.5]], g=[[2.0, 1.5], [2.0, 1.5]])),
]


def materialise(solver, kernel, inputs, stop, dtype=np.float64, rows=2, cols=2):
    """numpy arrays for every argument of the kernel, in C-ABI order"""
    arrays = {}
    for name, kind in KERNELS[solver][kernel]:
        if kind in "Vv":
            arrays[name] = np.full((rows, cols), float(inputs.get(name, 0.0)), dtype=dtype)
        elif kind in "Ss":
            arrays[name] = np.array(inputs.get(name, [0.0] * cols), dtype=dtype)
        elif kind == "U":
            arrays[name] = np.full(cols, 7, dtype=np.uint64)
        else:
            arrays[name] = np.array(stop, dtype=np.uint8)
    return arrays


def check(arrays, expected):
    for name, want in expected.items():
        got = arrays[name]
        want = np.broadcast_to(np.asarray(want, dtype=got.dtype), got.shape)
        assert np.array_equal(got, want), (name, got, want)


def random_case(solver, kernel, rows, cols, dtype, seed, zero_col=None, stopped_col=None):
    """seeded operands for the equivalence tests; optionally one column whose scalars are
    all zero (division-by-zero branches) and one stopped column"""
    rng = np.random.default_rng(seed)
    arrays = {}
    for name, kind in KERNELS[solver][kernel]:
        if kind in "Vv":
            arrays[name] = rng.uniform(-1, 1, (rows, cols)).astype(dtype)
        elif kind in "Ss":
            arrays[name] = (rng.uniform(0.5, 2.0, cols) * rng.choice([-1, 1], cols)).astype(dtype)
            if solver == "minres" and name == "beta":
                arrays[name] = np.abs(arrays[name])          # its square root is taken
            if zero_col is not None and zero_col < cols:
                arrays[name][zero_col] = 0
        elif kind == "U":
            arrays[name] = np.full(cols, 7, dtype=np.uint64)
        else:
            st = np.zeros(cols, dtype=np.uint8)
            if stopped_col is not None and stopped_col < cols:
                st[stopped_col] = STOPPED
            arrays[name] = st
    return arrays


# reference/test/solver/chebyshev_kernels.cpp:41-52 (fixture: alpha 0.5, beta 0.25) and
# :66-81 (KernelInitUpdate), :84-108 (KernelUpdate); all values are exact in binary
CHEB_INNER = [[0.5, 0.125, -0.125], [0.25, 0.5, -1.0], [1.5, -0.25, 1.5]]
CHEB_UPDATE = [[1.0, 0.0, -0.5], [0.5, 1.0, -1.0], [-1.5, -0.5, 1.0]]
CHEB_OUTPUT = [[-1.0, 0.5, -0.0], [0.75, 0.25, -1.25], [1.0, -1.25, 3.0]]


def check_chebyshev(kernel, inner, update, output):
    if kernel == "init_update":
        assert np.array_equal(update, np.array(CHEB_INNER, dtype=update.dtype))
        want = [[-0.75, 0.5625, -0.0625], [0.875, 0.5, -1.75], [1.75, -1.375, 3.75]]
    else:
        val = [[0.75, 0.125, -0.25], [0.375, 0.75, -1.25], [1.125, -0.375, 1.75]]
        assert np.array_equal(inner, np.array(val, dtype=inner.dtype))
        assert np.array_equal(update, inner)
        want = [[-0.625, 0.5625, -0.125], [0.9375, 0.625, -1.875], [1.5625, -1.4375, 3.875]]
    assert np.array_equal(output, np.array(want, dtype=output.dtype))
