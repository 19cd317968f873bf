"""Shared helpers for the parity tests."""
import numpy as np
import scipy.sparse as sp


def random_csr(rows, cols, density, seed, index_dtype=np.int32,
               dtype=np.float64, unsorted=False, empty_rows=()):
    rng = np.random.default_rng(seed)
    a = sp.random(rows, cols, density=density, random_state=rng, format="csr",
                  data_rvs=lambda n: rng.uniform(-1, 1, n))
    a.sort_indices()
    a = a.tolil()
    for r in empty_rows:
        a.rows[r] = []
        a.data[r] = []
    a = a.tocsr()
    a.sort_indices()
    row_ptrs = a.indptr.astype(index_dtype)
    cols_ = a.indices.astype(index_dtype)
    vals = a.data.astype(dtype)
    if unsorted:
        for r in range(rows):
            s, e = row_ptrs[r], row_ptrs[r + 1]
            perm = rng.permutation(e - s)
            cols_[s:e] = cols_[s:e][perm]
            vals[s:e] = vals[s:e][perm]
    return row_ptrs, cols_, vals


def rel_frobenius(a, b):
    """Ginkgo's GKO_ASSERT_MTX_NEAR metric
    (core/test/utils/assertions.hpp:275-307)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    num = np.sqrt(np.sum((a - b) ** 2))
    den = np.sqrt(max(np.sum(a ** 2), np.sum(b ** 2)))
    return 0.0 if num == 0 else num / (den if den > 0 else 1.0)


def record_perf(name, **values):
    """Timings and ratios the GPU tests OBSERVE are recorded, not asserted: a
    wall-clock bound that holds on one box fails on the next and takes every later test with it under -x.
    One JSON line per observation in gpurun_out/perf_records.jsonl; GKO_TEST_PERF=1 turns the bounds back on
    (the callers check `perf_asserts()`)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "perf_records.jsonl"), "a") as f:
            f.write(json.dumps({"name": name, **values}) + "\n")
    except OSError:
        pass


def perf_asserts():
    import os
    return os.environ.get("GKO_TEST_PERF", "0") == "1"
