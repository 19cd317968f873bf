"""dot / norm2 / squared_norm2 of one contiguous column in ONE launch (GKOC_TUNE_REDUCE_ONE_KERNEL: the
block that finishes last folds the partial sums, csrc/dense.hip reduce_flat_stage1<.., FOLD>) must give
the bits of the two-launch form - same threads, same strided sums, same tree - for every size, on
several streams at once, and a thousand times in a row (the fold reads sums other XCDs wrote)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KEY = 9   # GKOC_TUNE_REDUCE_ONE_KERNEL


def _set(v):
    from ginkgo_amd._lib import call
    call("gkoc_tune_set", C.c_int(KEY), C.c_int64(v))


@pytest.fixture
def one_kernel():
    from ginkgo_amd._lib import lib
    old = C.c_int64(0)
    lib().gkoc_tune_get(C.c_int(KEY), C.byref(old))
    yield
    _set(old.value)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_bits_of_the_two_launch_form(gexec, one_kernel, dtype):
    import ginkgo_amd as g
    rng = np.random.default_rng(5)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    for n in (1, 100, 4096, 4097, 65536, 1 << 20, (1 << 22) + 3, 16777216 // 8):
        x = g.Dense.from_numpy(gexec, rng.uniform(-1, 1, n).astype(dtype))
        y = g.Dense.from_numpy(gexec, rng.uniform(-1, 1, n).astype(dtype))
        got = {}
        for mode in (0, 1):
            _set(mode)
            r = [g.Dense.create(gexec, (1, 1), tdt) for _ in range(3)]
            x.compute_dot(y, r[0])
            x.compute_norm2(r[1])
            x.compute_squared_norm2(r[2])
            torch.cuda.synchronize()
            got[mode] = [t.to_numpy().view(np.uint8).copy() for t in r]
        for a, b in zip(got[0], got[1]):
            assert np.array_equal(a, b), n
        # an unaligned view takes the scalar-load path of the same kernel
        xv = g.Dense(gexec, x.values[1:]) if n > 4 else None
        if xv is not None:
            yv = g.Dense(gexec, y.values[1:])
            out = []
            for mode in (0, 1):
                _set(mode)
                r = g.Dense.create(gexec, (1, 1), tdt)
                xv.compute_dot(yv, r)
                out.append(r.to_numpy().view(np.uint8).copy())
            assert np.array_equal(out[0], out[1]), n


def test_a_thousand_times_and_on_two_streams(gexec, one_kernel):
    """the same sum every time; two streams' reductions in flight together use a ticket word each"""
    import ginkgo_amd as g
    from ginkgo_amd._lib import call
    rng = np.random.default_rng(9)
    n = 2_100_000                     # a rank's share of 256^3 / 8: 514 partial sums
    x = g.Dense.from_numpy(gexec, rng.uniform(-1, 1, n))
    y = g.Dense.from_numpy(gexec, rng.uniform(-1, 1, n))
    _set(0)
    ref = g.Dense.create(gexec, (1, 1))
    x.compute_dot(y, ref)
    want = ref.to_numpy().copy()
    _set(1)
    outs = gexec.zeros((1000,), torch.float64)
    w = x._work(1)
    for k in range(1000):
        call("gkoc_dense_compute_dot_f64", gexec.stream, n, 1, x.values, 1, y.values, 1, outs[k:k + 1], w,
             C.c_size_t(w.numel()))
    torch.cuda.synchronize()
    assert np.array_equal(outs.cpu().numpy(), np.full(1000, want[0, 0]))
    side = torch.cuda.Stream()
    w2 = gexec.alloc((w.numel(),), w.dtype)
    outs2 = gexec.zeros((200,), torch.float64)
    outs.zero_()
    torch.cuda.synchronize()
    for k in range(200):
        call("gkoc_dense_compute_dot_f64", gexec.stream, n, 1, x.values, 1, y.values, 1, outs[k:k + 1], w,
             C.c_size_t(w.numel()))
        call("gkoc_dense_compute_dot_f64", C.c_void_p(side.cuda_stream), n, 1, y.values, 1, x.values, 1,
             outs2[k:k + 1], w2, C.c_size_t(w2.numel()))
    torch.cuda.synchronize()
    assert np.array_equal(outs.cpu().numpy()[:200], np.full(200, want[0, 0]))
    assert np.array_equal(outs2.cpu().numpy(), np.full(200, want[0, 0]))
