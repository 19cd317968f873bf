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

from util import perf_asserts, record_perf

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
    # What the survey FINDS is the box's business (how many classes lie within the walk, what a handle costs:
    # another process's memory is cleared by the driver at about 30 ms per GiB): recorded, and asserted
    # only as "what was found is used as designed".  Timings and ratios are recorded too (tests/util.py
    # record_perf; GKO_TEST_PERF=1 brings the bounds back for a dedicated run on a quiet box).
    record_perf("arena_classes_spmv_256", three=r3, two=r2, one=r1, hipmalloc=r0)
    assert 1 <= r3["classes_found"] <= 3, r3
    c = r3["class_of"]
    if r3["classes_found"] == 3:       # values, indices, vectors apart
        assert len({c["values"], c["col_idxs"], c["y"]}) == 3 and c["x"] == c["y"], c
    assert r2["classes_found"] <= 2 and r1["classes_found"] == 1 and r0["mode"] == 0
    if r2["classes_found"] == 2:       # matrix | vectors
        c = r2["class_of"]
        assert c["values"] == c["col_idxs"] != c["y"] and c["x"] == c["y"], c
    assert r3["granules_classified"] <= 80, r3          # the search gallops: few walked granules are probed
    if perf_asserts():
        assert r3["search_ms"] < 1000 + 100 * r3["granules_walked"], r3
        # two classes cost a few per cent (measured 2 %), one class the 11 % of DESIGN.md 3.2
        assert r2["ms"] <= 1.05 * r3["ms"], (r2["ms"], r3["ms"])
        assert r1["ms"] <= 1.22 * r3["ms"], (r1["ms"], r3["ms"])
        # never slower than the reference's one hipMalloc per array (hip/base/executor.hip.cpp:95-112)
        assert r3["ms"] <= 1.02 * r0["ms"], (r3["ms"], r0["ms"])
        assert r2["ms"] <= 1.05 * r0["ms"], (r2["ms"], r0["ms"])


def test_search_bounded_by_a_walk_limit_settles_for_what_it_found():
    """GKOC_ARENA_MAX_WALK=2: the survey may create two granules - whatever it found, the
    allocator works, the result is the same"""
    r = _child(grid=64, warm=2, steps=2, GKOC_ARENA_MAX_WALK=2)
    ref = _child(grid=64, warm=2, steps=2)
    assert r["digest"] == ref["digest"]
    assert 1 <= r["classes_found"] <= 3 and r["granules_walked"] <= 4, r
