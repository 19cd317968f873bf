"""tools/gko_benchmark.py writes what the reference's benchmark/spmv and benchmark/solver
write: same input objects, same keys and nesting as benchmark/test/reference/
spmv.simple.stdout and solver.simple.stdout (the 7pt case of size 100 = 125 rows, 725
nonzeros, coo storage 11600 bytes, cg converging in 7 iterations there)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INPUT = '[{"size": 100, "stencil": "7pt"}]'


def _run(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gko_benchmark.py"), *args],
                       input=INPUT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout), p.stderr


def test_spmv_schema():
    out, err = _run("spmv", "-formats", "coo,csr,ell,sellp,hybrid")
    assert "Matrix is of size (125, 125), 725" in err          # spmv.simple.stderr
    case = out[0]
    assert set(case) == {"size", "stencil", "spmv", "rows", "cols", "nonzeros", "optimal"}
    assert (case["size"], case["stencil"], case["rows"], case["cols"], case["nonzeros"]) == (100, "7pt", 125, 125, 725)
    assert set(case["spmv"]) == {"coo", "csr", "ell", "sellp", "hybrid"}
    for fmt, res in case["spmv"].items():
        assert set(res) == {"storage", "max_relative_norm2", "time", "repetitions", "completed"}, fmt
        assert res["completed"] is True and res["repetitions"] == 10 and res["time"] > 0
        assert res["max_relative_norm2"] < 1e-14
    assert case["spmv"]["coo"]["storage"] == 11600             # spmv.simple.stdout
    assert case["optimal"]["spmv"] in case["spmv"]


def test_solver_schema():
    out, _ = _run("solver", "-solvers", "cg,bicgstab", "-preconditioners", "none,jacobi", "-rel_res_goal", "1e-6")
    case = out[0]
    assert set(case) == {"size", "stencil", "optimal", "solver", "rows", "cols"}
    assert set(case["solver"]) == {"cg", "cg-jacobi", "bicgstab", "bicgstab-jacobi"}
    for name, res in case["solver"].items():
        assert set(res) == {"recurrent_residuals", "true_residuals", "implicit_residuals", "iteration_timestamps",
                            "rhs_norm", "generate", "apply", "preconditioner", "residual_norm", "repetitions",
                            "completed"}, name
        assert res["completed"] is True
        assert set(res["apply"]) == {"components", "iterations", "time"}
        assert res["residual_norm"] <= 1e-6 * res["rhs_norm"] * 1.01
    assert case["solver"]["cg"]["apply"]["iterations"] == 7     # solver.simple.stdout
