"""Drop-in acceptance on the GPU box: the UNMODIFIED Ginkgo core
(oracle/_ref/lib/libginkgo.so) on gko::HipExecutor, with libginkgo_hip.so =
ginkgo_amd/gko_binding (shim) + libgko_cdna4.so.  Runs tests/dropin/dropin_test.cpp
and Ginkgo's own examples/preconditioned-solver (golden output
examples/preconditioned-solver/doc/results.dox: residual 4.82005e-08).
The binaries are built in the container by oracle/build_dropin.py."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROP = os.path.join(ROOT, "oracle", "_ref", "dropin")

pytestmark = pytest.mark.gpu


def _need(name):
    path = os.path.join(DROP, name)
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (run oracle/build_dropin.py where /root/reference exists)")
    return path


def test_dropin_program():
    exe = _need("dropin_test")
    p = subprocess.run([exe, "24"], capture_output=True, text=True, timeout=600, cwd=DROP)
    print(p.stdout[-4000:])
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert "DROPIN OK" in p.stdout
    assert "bit-identical" in p.stdout and "FAILED" not in p.stdout
    # the checks of the later rounds ran (none of them is skipped silently): block-wise / adaptive
    # block-Jacobi for the three value types next to double, 8 requests x 3 checks each and case
    for vt, cases in (("float", 2), ("complex<double>", 2), ("complex<float>", 1)):
        got = len(re.findall(r"^ok: Jacobi<" + re.escape(vt) + "> ", p.stdout, re.M))
        assert got == 24 * cases, (vt, got)


def _numbers(block):
    return [float(x) for x in re.findall(r"^[-+]?[0-9][-+0-9.eE]*$", block, re.M)]


def test_ginkgo_example_preconditioned_solver():
    exe = _need("preconditioned-solver")
    out = {}
    for ex in ("reference", "hip"):
        p = subprocess.run([exe, ex], capture_output=True, text=True, timeout=600, cwd=DROP)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        out[ex] = p.stdout
    # golden output (doc/results.dox): 19 solution values, then the residual norm 4.82005e-08
    sol, res = {}, {}
    for k, v in out.items():
        head, tail = v.split("Residual norm")
        sol[k] = _numbers(head)[-19:]
        res[k] = _numbers(tail)[-1]
    assert abs(res["reference"] - 4.82005e-08) < 1e-12
    assert abs(res["hip"] - res["reference"]) < 1e-12
    assert len(sol["hip"]) == len(sol["reference"]) == 19
    assert max(abs(a - b) for a, b in zip(sol["hip"], sol["reference"])) < 1e-5


def test_ginkgo_api_benchmark_program_runs():
    """tests/dropin/dropin_bench.cpp: Csr / Ell / Sellp apply and Cg + Jacobi(8) of the
    unmodified core, timed with gko::Timer (small grid here; 256^3 numbers in
    profiles/r01_ginkgo_api_bench_on_this_backend.txt)"""
    exe = _need("dropin_bench")
    p = subprocess.run([exe, "48", "5", "20"], capture_output=True, text=True, timeout=600, cwd=DROP)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    for key in ("gko::matrix::Csr::apply", "gko::matrix::Ell::apply", "gko::matrix::Sellp::apply",
                "gko::solver::Cg + Jacobi(8)  20 iterations"):
        assert key in p.stdout, p.stdout


def test_mixed_precision_core_flavor():
    """tests/dropin/mixed_test.cpp on the core built with GINKGO_MIXED_PRECISION
    (oracle/build_ref_mixed.py): every (matrix, input, output) value-type triple of csr / ell
    spmv + advanced_spmv (8 real + 8 complex, int32 / int64, 1 and 3 columns) and the mixed
    dense::row_gather pairs, hip against ReferenceExecutor in one process - real triples bit for bit"""
    exe = _need("mixed_test")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600, cwd=DROP)
    print(p.stdout[-4000:])
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert "MIXED OK" in p.stdout and "FAILED" not in p.stdout
    n = int(re.search(r"(\d+) checks, 0 failed", p.stdout).group(1))
    assert n >= 2 * 2 * 2 * 8 * 2 * 3 + 8      # flavors x index types x columns x triples x formats x cases
