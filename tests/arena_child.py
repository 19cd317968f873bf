"""Child process of tests/test_arena_classes_gpu.py: the 27-pt grid^3 CSR SpMV (BASELINE configs[1])
in a fresh process under whatever GKOC_ARENA* the parent set; prints one JSON object with the
kernel time (HIP events on the launch stream, `warm` untimed launches first: the chip needs
20-30 ms to reach its clocks), the memory classes of the operands and a digest of y."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    grid, warm, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    import numpy as np
    import torch
    import ginkgo_amd as g
    ex = g.Cdna4Executor.create(0)
    a = g.stencil_csr(ex, 3, grid)
    n = grid ** 3
    x = g.Dense.from_numpy(ex, np.random.default_rng(42).uniform(-1, 1, n))
    y = g.Dense.create(ex, (n, 1))
    for _ in range(warm):
        a.apply(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        a.apply(x, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    info = ex.arena_info()
    print(json.dumps({
        "ms": ms, "digest": hashlib.sha256(y.values.cpu().numpy().tobytes()).hexdigest(),
        "classes_found": info["num_classes"], "granules_walked": info["granules_walked"],
        "granules_classified": info["granules_classified"], "search_ms": info["search_ns"] / 1e6,
        "probe_retries": info["probe_retries"], "mode": info["mode"],
        "class_of": {"values": ex.memory_class(a.values), "col_idxs": ex.memory_class(a.col_idxs),
                     "x": ex.memory_class(x.values), "y": ex.memory_class(y.values)}}))


if __name__ == "__main__":
    main()
