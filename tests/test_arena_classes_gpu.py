"""The headline must not depend on the allocator's luck: BENCH_r03 fell
from 73.8 to 70.6 % of the HBM peak because the arena's search met only two of the three memory
classes of the device and the output vector ended up next to the column indices.  Here that
outcome is FORCED (GKOC_ARENA_MAX_CLASSES) in fresh processes and the 256^3 SpMV compared with the
three-class run: bit-identical, and within a few per cent - with two classes the matrix arrays
share one and everything kernels write gets the other.  The reference's behaviour - one hipMalloc
per array (hip/base/executor.hip.cpp:95-112), GKOC_ARENA=0 - is what the arena must never lose to."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(grid=256, warm=25, steps=20, **env):
    e = dict(os.environ)
    for k in list(e):
        if k.startswith("GKOC_ARENA"):
            del e[k]
    e.update({k: str(v) for k, v in env.items()})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "arena_child.py"), str(grid), str(warm),
                        str(steps)], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert p.returncode == 0, p.stdout + p.stderr
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def test_spmv_256_with_three_two_and_one_memory_classes():
    r3 = _child()
    r2 = _child(GKOC_ARENA_MAX_CLASSES=2)
    r1 = _child(GKOC_ARENA_MAX_CLASSES=1)
    r0 = _child(GKOC_ARENA=0)
    for r in (r3, r2, r1, r0):
        print(json.dumps(r))
    assert r3["digest"] == r2["digest"] == r1["digest"] == r0["digest"], "results depend on the placement"
    # three classes: values, indices, vectors apart
    assert r3["classes_found"] == 3, r3
    c = r3["class_of"]
    assert len({c["values"], c["col_idxs"], c["y"]}) == 3 and c["x"] == c["y"], c
    # two classes: matrix | vectors
    assert r2["classes_found"] == 2, r2
    c = r2["class_of"]
    assert c["values"] == c["col_idxs"] != c["y"] and c["x"] == c["y"], c
    assert r1["classes_found"] == 1 and r0["mode"] == 0
    # the search gallops: few of the walked granules are mapped and probed.  What it costs depends on the
    # box: a handle of memory another process has used is cleared by the driver when it is created (about
    # 30 ms per GiB; 3 ms on clean memory), and how many handles lie in front of the third class is the
    # driver's business (8 - 151 seen) - so the bound is per handle, not absolute
    assert r3["granules_classified"] <= 80, r3
    assert r3["search_ms"] < 1000 + 100 * r3["granules_walked"], r3
    # two classes cost a few per cent (values and indices in one class: measured 2 %), not the
    # 6 % of BENCH_r03 (y next to the indices); one class is the 11 % of DESIGN.md 3.2
    # (measured + 3.3 % and + 15 %; one process each, so the margins also hold the run-to-run spread)
    assert r2["ms"] <= 1.05 * r3["ms"], (r2["ms"], r3["ms"])
    assert r1["ms"] <= 1.22 * r3["ms"], (r1["ms"], r3["ms"])
    # never slower than the reference's one hipMalloc per array (whose lottery is kind in a fresh
    # process: 0.947 - 0.950 ms measured against 0.940 - 0.941 with three classes)
    assert r3["ms"] <= 1.02 * r0["ms"], (r3["ms"], r0["ms"])
    assert r2["ms"] <= 1.05 * r0["ms"], (r2["ms"], r0["ms"])


def test_search_bounded_by_a_walk_limit_settles_for_what_it_found():
    """GKOC_ARENA_MAX_WALK=2: the survey may create two granules - whatever it found, the
    allocator works, the result is the same"""
    r = _child(grid=64, warm=2, steps=2, GKOC_ARENA_MAX_WALK=2)
    ref = _child(grid=64, warm=2, steps=2)
    assert r["digest"] == ref["digest"]
    assert 1 <= r["classes_found"] <= 3 and r["granules_walked"] <= 4, r
