"""tests/golden/jacobi_types.npz (the reference's block-Jacobi decisions and products for float and
complex values, consumed on the GPU by tests/test_jacobi_types_gpu.py): complete, and - where the
reference sources and oracle/_ref exist, i.e. in the build container - equal to what the live
reference computes now (tests/golden/make_jacobi_types_golden.py --check)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "jacobi_types.npz")


def test_the_fixture_has_every_case_and_every_storage_type():
    g = np.load(GOLD)
    cases = sorted({tuple(k.split("/")[:3]) for k in g.files if k.count("/") == 3})
    assert len(cases) == 40 and {c[0] for c in cases} == {"f32", "c64", "c128"}
    kinds = set()
    for vt, bs, tag in cases:
        prec, ptrs = g[f"{vt}/{bs}/{tag}/prec"], g[f"{vt}/{bs}/{tag}/block_ptrs"]
        assert len(prec) == len(ptrs) - 1 and ptrs[0] == 0 and ptrs[-1] == len(g[f"{vt}/row_ptrs"]) - 1
        assert (np.diff(ptrs) >= 1).all() and (np.diff(ptrs) <= int(bs)).all()
        # blocks of one storage group (64 / next power of two of max_block_size of them) share a precision
        p2 = 1
        while p2 < int(bs):
            p2 *= 2
        gs = 64 // p2
        for first in range(0, len(prec), gs):
            assert len(set(prec[first:first + gs].tolist())) == 1, (vt, bs, tag, first)
        kinds |= set(prec.tolist())
    assert kinds == {0x00, 0x01, 0x02, 0x10, 0x11, 0x20}


def test_the_fixture_equals_the_live_reference():
    ref = os.environ.get("GKO_REFERENCE_DIR", "/root/reference")
    if not (os.path.isdir(os.path.join(ref, "include", "ginkgo"))
            and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "lib", "libginkgo.so"))):
        pytest.skip("needs the reference sources and oracle/_ref (the build container)")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_jacobi_types_golden.py"),
                        "--check"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "fixture == live reference" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def _cases():
    g = np.load(GOLD)
    return sorted({tuple(k.split("/")[:3]) for k in g.files if k.count("/") == 3})


@pytest.mark.parametrize("vt,max_bs,tag", _cases())
def test_the_oracle_restatement_is_pinned_by_the_fixture(vt, max_bs, tag):
    """oracle/gko_oracle_jacobi_types.inc against the reference's vectors, BIT FOR BIT: the blocks' precisions,
    the condition numbers, M b, M^T b and M^H b for the three value types (the C restatement runs the
    reference's operations in the reference's order, complex products and quotients through the same
    libgcc routines std::complex uses)"""
    from oracle import gko_oracle as o
    g = np.load(GOLD)
    gold = lambda k: g[f"{vt}/{max_bs}/{tag}/{k}"]      # noqa: E731
    rp, ci, vals, b = (g[f"{vt}/{k}"] for k in ("row_ptrs", "col_idxs", "values", "b"))
    ptrs = gold("block_ptrs")
    nb = len(ptrs) - 1
    scheme = o.jacobi_storage_scheme(int(max_bs))
    blocks, prec, cond = o.jacobi_generate_adaptive_t(rp, ci, vals, nb, scheme, ptrs, float(gold("accuracy")[0]),
                                                      gold("request"))
    assert np.array_equal(prec, gold("prec"))
    assert cond.astype(np.float64).tobytes() == gold("cond").tobytes()
    x = o.jacobi_apply_adaptive_t(nb, scheme, ptrs, blocks, prec, b)
    assert x.tobytes() == gold("x").tobytes()
    for conj, key in ((False, "xt"), (True, "xh")):
        t = o.jacobi_transpose_adaptive_t(nb, scheme, ptrs, blocks, prec, conj)
        assert o.jacobi_apply_adaptive_t(nb, scheme, ptrs, t, prec, b).tobytes() == gold(key).tobytes(), key
