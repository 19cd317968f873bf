"""Placement by the backend's allocator when nothing states a role - through the UNMODIFIED Ginkgo
API, where gko::HipExecutor::raw_alloc is all the allocator sees (csrc/arena.hip class_for_role,
gkoc_arena_note_vector; tests/dropin/arena_roles_test.cpp).  Three allocation orders that the
size-only rule of round 2 got wrong: vectors allocated before the matrix, a Krylov basis larger than
the matrix' values, a 5-point matrix.  Asserted: no array that kernels write shares a memory class
with the values or column indices; printed: the kernel times next to GKOC_ARENA=0 (one hipMalloc
per array, the reference's behaviour)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "dropin", "arena_roles_test")


def _run(scenario, arena=None):
    env = dict(os.environ)
    if arena is not None:
        env["GKOC_ARENA"] = str(arena)
    p = subprocess.run([EXE, scenario], capture_output=True, text=True, timeout=900, env=env,
                       cwd=os.path.dirname(EXE))
    return p.returncode, p.stdout + p.stderr


@pytest.mark.parametrize("scenario", ["vectors-first", "gmres-basis", "five-point"])
def test_written_arrays_never_share_a_class_with_matrix_arrays(scenario):
    if not os.path.exists(EXE):
        pytest.skip("oracle/build_dropin.py has not been run (needs /root/reference)")
    rc, out = _run(scenario)
    print(out)
    assert rc == 0 and "MISPLACED" not in out, out
    assert out.count("ok:") >= (2 if scenario == "gmres-basis" else 1), out
    m = re.search(r"memory classes: values (-?\d+), col_idxs (-?\d+), row_ptrs (-?\d+), b (-?\d+), x (-?\d+)", out)
    cv, cc, cr, cb, cx = map(int, m.groups())
    assert min(cv, cc, cb, cx) >= 0, "the arrays do not live in the arena's class regions"
    # for the record: the same program with one hipMalloc per array
    rc0, out0 = _run(scenario, arena=0)
    t = lambda o, key: float(re.search(key + r"\s+([\d.]+) ms", o).group(1))
    print(f"{scenario}: Csr::apply {t(out, 'Csr::apply'):.4f} ms with the arena, "
          f"{t(out0, 'Csr::apply'):.4f} ms with GKOC_ARENA=0")
    if scenario == "gmres-basis":
        k = r"30 iterations,"
        print(f"{scenario}: Gmres(30) {t(out, k):.4f} ms/iteration with the arena, "
              f"{t(out0, k):.4f} with GKOC_ARENA=0")
