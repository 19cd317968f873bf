"""The C++ host-side example (examples/native_cg.cpp) drives CG + block-Jacobi(8)
through the C ABI only.  Compared with the oracle's CG on the same matrix."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "native_cg")


def _run(*args):
    out = subprocess.run([EXE, *map(str, args)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.skipif(not os.path.exists(EXE), reason="examples/native_cg not built (run build())")
def test_native_cg_matches_oracle(oracle):
    grid = 24
    rp, ci, v = oracle.stencil_csr(3, grid)
    n = grid ** 3
    xo, iters, _ = oracle.cg_solve(rp, ci, v, np.ones(n), max_iters=1000, reduction=1e-10,
                                   precond="block", max_block_size=8)
    res = {}
    for mode, lag in (("plain", 0), ("fused", 0), ("fused", 4), ("fused", 7), ("graph", 4)):
        r = _run(grid, 1000, 1e-10, mode, lag)
        assert r["converged"] and abs(r["iterations"] - iters) <= 1
        assert r["true_rel_residual"] <= 1.01e-10
        assert abs(r["x_sum"] - xo.sum()) <= 1e-8 * abs(xo.sum())
        res[(mode, lag)] = r
    # reading the criterion late changes nothing, bit for bit
    assert res[("fused", 0)]["x_sum"] == res[("fused", 4)]["x_sum"] == res[("fused", 7)]["x_sum"]
    assert res[("fused", 0)]["iterations"] == res[("fused", 4)]["iterations"]
    # ... and so does replaying two captured iterations as a hipGraph
    assert res[("graph", 4)]["x_sum"] == res[("fused", 0)]["x_sum"]
    assert res[("graph", 4)]["iterations"] == res[("fused", 0)]["iterations"]
    # iteration limit
    for mode in ("fused", "graph"):
        for cap in (5, 6):
            r = _run(grid, cap, 1e-30, mode, 4)
            assert r["iterations"] == cap and not r["converged"]
