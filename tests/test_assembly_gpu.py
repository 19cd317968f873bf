"""GPU parity of the device-side assembly of device_matrix_data:
components::{aos_to_soa, soa_to_aos, sort_row_major, remove_zeros, sum_duplicates} and
Csr::read(device_matrix_data), through the C ABI, against the oracle and the golden
arrays of the reference.

Mirrors test/base/device_matrix_data_kernels.cpp (100 x 200, 1000 random entries + 1000
explicit zeros, shuffled, + 1000 duplicated locations; SortsRowMajor, RemovesZeros,
DoesntRemoveZerosIfThereAreNone, SumsDuplicates, DoesntSumDuplicatesIfThereAreNone,
CreatesFromHost, CopiesToHost).  Bar: every array bit-identical - the sums of the
duplicate runs included (the reference test allows 2 r there; runs are summed in storage
order from 0 here, like the reference's loop)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COMBOS = [(np.float64, np.int32), (np.float64, np.int64), (np.float32, np.int32), (np.float32, np.int64)]


def _md(g, gexec, size, rows, cols, vals):
    return g.DeviceMatrixData(gexec, size, gexec.to_device(rows), gexec.to_device(cols), gexec.to_device(vals))


def _get(md):
    return md.row_idxs.cpu().numpy(), md.col_idxs.cpu().numpy(), md.values.cpu().numpy()


def _same(got, want):
    for a, b in zip(got, want):
        assert a.dtype == b.dtype and a.tobytes() == b.tobytes()


def _fixture(vdt, idt, seed=82754):
    """the data set of test/base/device_matrix_data_kernels.cpp:30-76"""
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, 100, 2000).astype(idt)
    cols = rng.integers(0, 200, 2000).astype(idt)
    vals = rng.uniform(1, 2, 2000).astype(vdt)
    vals[1000:] = 0
    _, first = np.unique(rows.astype(np.int64) * 200 + cols, return_index=True)
    p = rng.permutation(first)
    host = rows[p], cols[p], vals[p]
    loc = rng.integers(0, len(p), 1000)
    dup = tuple(np.concatenate((a, b)) for a, b in
                zip(host, (host[0][loc], host[1][loc], rng.uniform(1, 2, 1000).astype(vdt))))
    return host, dup


def test_golden_assembly(gexec):
    import ginkgo_amd as g
    gd = np.load(os.path.join(GOLD, "assembly.npz"))
    size = tuple(int(t) for t in gd["shape"])
    args = (gd["rows"], gd["cols"], gd["vals"])
    _same(_get(_md(g, gexec, size, *args).sort_row_major()), [gd["sort_row_major_" + k] for k in ("rows", "cols", "vals")])
    _same(_get(_md(g, gexec, size, *args).remove_zeros()), [gd["remove_zeros_" + k] for k in ("rows", "cols", "vals")])
    _same(_get(_md(g, gexec, size, *args).sum_duplicates()), [gd["sum_duplicates_" + k] for k in ("rows", "cols", "vals")])
    csr = g.Csr.read(_md(g, gexec, size, *args).sort_row_major())
    _same((csr.row_ptrs.cpu().numpy(), csr.col_idxs.cpu().numpy(), csr.values.cpu().numpy()),
          [gd["csr_" + k] for k in ("rows", "cols", "vals")])


@pytest.mark.parametrize("vdt,idt", COMBOS)
def test_reference_test_fixture(gexec, oracle, vdt, idt):
    import ginkgo_amd as g
    host, dup = _fixture(vdt, idt)
    srt = oracle.md_sort_row_major(*host)
    _same(_get(_md(g, gexec, (100, 200), *host).sort_row_major()), srt)                      # SortsRowMajor
    nz = oracle.md_remove_zeros(*host)
    assert len(nz[2]) < len(host[2])
    _same(_get(_md(g, gexec, (100, 200), *host).remove_zeros()), nz)                         # RemovesZeros
    md = _md(g, gexec, (100, 200), *nz)                                                      # DoesntRemoveZerosIfThereAreNone
    before = (md.row_idxs.data_ptr(), md.col_idxs.data_ptr(), md.values.data_ptr())
    md.remove_zeros()
    assert before == (md.row_idxs.data_ptr(), md.col_idxs.data_ptr(), md.values.data_ptr())
    _same(_get(md), nz)
    want = oracle.md_sum_duplicates(*oracle.md_sort_row_major(*dup))                        # SumsDuplicates
    assert len(want[2]) == len(host[2])
    _same(_get(_md(g, gexec, (100, 200), *dup).sum_duplicates()), want)
    md = _md(g, gexec, (100, 200), *host)                                                    # DoesntSumDuplicatesIfThereAreNone
    before = (md.row_idxs.data_ptr(), md.col_idxs.data_ptr(), md.values.data_ptr())
    md.sum_duplicates()
    assert before == (md.row_idxs.data_ptr(), md.col_idxs.data_ptr(), md.values.data_ptr())
    _same(_get(md), srt)


@pytest.mark.parametrize("vdt,idt", COMBOS)
@pytest.mark.parametrize("nnz,n_rows,n_cols", [(0, 5, 5), (1, 5, 5), (2, 1, 1), (63, 4, 4), (257, 9, 3),
                                                (4097, 50, 50), (70000, 300, 7), (300000, 100000, 90000)])
def test_random_assembly(gexec, oracle, vdt, idt, nnz, n_rows, n_cols):
    import ginkgo_amd as g
    rng = np.random.default_rng(nnz + 7)
    rows = rng.integers(0, n_rows, nnz).astype(idt)
    cols = rng.integers(0, n_cols, nnz).astype(idt)
    vals = rng.standard_normal(nnz).astype(vdt)
    vals[rng.random(nnz) < 0.2] = 0
    vals[rng.random(nnz) < 0.05] = -0.0
    srt = oracle.md_sort_row_major(rows, cols, vals)
    _same(_get(_md(g, gexec, (n_rows, n_cols), rows, cols, vals).sort_row_major()), srt)
    _same(_get(_md(g, gexec, (n_rows, n_cols), rows, cols, vals).remove_zeros()), oracle.md_remove_zeros(rows, cols, vals))
    _same(_get(_md(g, gexec, (n_rows, n_cols), rows, cols, vals).sum_duplicates()), oracle.md_sum_duplicates(*srt))


def test_edge_values(gexec, oracle):
    import ginkgo_amd as g
    i = np.arange(6, dtype=np.int32)
    vals = np.array([0.0, -0.0, np.nan, np.inf, 5e-324, 1.0])
    got = _get(_md(g, gexec, (6, 6), i, i, vals).remove_zeros())
    _same(got, oracle.md_remove_zeros(i, i, vals))
    assert len(got[2]) == 4 and np.isnan(got[2][0])
    # all zeros -> empty; all the same location -> one entry
    z = _md(g, gexec, (6, 6), i, i, np.zeros(6)).remove_zeros()
    assert z.get_num_stored_elements() == 0 and z.row_idxs.numel() == 0
    one = np.zeros(1000, np.int32)
    v = np.random.default_rng(0).standard_normal(1000)
    _same(_get(_md(g, gexec, (1, 1), one, one, v).sum_duplicates()), oracle.md_sum_duplicates(one, one, v))
    # -0 survives sum_duplicates only when nothing is merged
    r = np.array([0, 1, 1], np.int32)
    _same(_get(_md(g, gexec, (2, 2), r, r, np.array([-0.0, 1.0, 2.0])).sum_duplicates()),
          oracle.md_sum_duplicates(r, r, np.array([-0.0, 1.0, 2.0])))
    r = np.array([1, 0], np.int32)
    got = _get(_md(g, gexec, (2, 2), r, r, np.array([1.0, -0.0])).sum_duplicates())
    assert got[0].tolist() == [0, 1] and np.signbit(got[2][0])
    # negative keys sort like signed integers (the reference compares index_type values)
    r = np.array([3, -1, 2, -5], np.int32)
    _same(_get(_md(g, gexec, (4, 4), r, r[::-1].copy(), np.arange(4.0)).sort_row_major()),
          oracle.md_sort_row_major(r, r[::-1].copy(), np.arange(4.0)))


@pytest.mark.parametrize("vdt,idt", COMBOS)
def test_host_round_trip(gexec, vdt, idt):
    """CreatesFromHost / CopiesToHost: matrix_data entries <-> device arrays"""
    import ginkgo_amd as g
    host, _ = _fixture(vdt, idt, seed=5)
    dt = g.entry_dtype(vdt, idt)
    entries = np.zeros(len(host[2]), dt)
    entries["row"], entries["column"], entries["value"] = host
    md = g.DeviceMatrixData.create_from_host(gexec, (100, 200), entries)
    _same(_get(md), host)
    back = md.copy_to_host()
    assert back.dtype == dt
    for k, a in zip(("row", "column", "value"), host):
        assert back[k].tobytes() == a.tobytes()


def test_assemble_stencil_on_device(gexec):
    """size-independent properties at 128^3 (56 M entries): shuffled 27-point stencil
    triplets + an explicit zero per row + every entry split in two halves ->
    sum_duplicates + remove_zeros give back the stencil's CSR arrays exactly, and
    sorting again changes nothing"""
    import ginkgo_amd as g
    a = g.stencil_csr(gexec, 3, 128)
    n, nnz = a.size[0], a.get_num_stored_elements()
    coo = a.convert_to_coo()
    gen = torch.Generator(device=coo.values.device).manual_seed(1)
    rows = torch.cat((coo.row_idxs, coo.row_idxs, torch.arange(n, dtype=torch.int32, device=coo.values.device)))
    cols = torch.cat((coo.col_idxs, coo.col_idxs, torch.full((n,), 7, dtype=torch.int32, device=coo.values.device)))
    vals = torch.cat((coo.values * 0.25, coo.values * 0.75, torch.zeros(n, dtype=torch.float64, device=coo.values.device)))
    p = torch.randperm(rows.numel(), generator=gen, device=rows.device)
    md = g.DeviceMatrixData(gexec, a.size, rows[p].contiguous(), cols[p].contiguous(), vals[p].contiguous())
    del rows, cols, vals, p
    md.sum_duplicates()
    md.remove_zeros()
    assert md.get_num_stored_elements() == nnz
    csr = g.Csr.read(md)
    assert torch.equal(csr.row_ptrs, a.row_ptrs) and torch.equal(csr.col_idxs, a.col_idxs)
    assert torch.equal(csr.values, a.values)        # 0.25 v + 0.75 v == v exactly for v = 26, -1
    before = tuple(t.clone() for t in (md.row_idxs, md.col_idxs, md.values))
    md.sort_row_major()
    assert all(torch.equal(x, y) for x, y in zip(before, (md.row_idxs, md.col_idxs, md.values)))
