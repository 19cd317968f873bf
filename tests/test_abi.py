"""CPU-side checks of the drop-in boundary: libgko_cdna4.so loads, exports
every symbol include/gko_cdna4.h declares, and fails loudly without a GPU."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gko_cdna4.h")


def declared_symbols():
    pre = subprocess.run(["gcc", "-E", "-P", HEADER], capture_output=True,
                         text=True, check=True).stdout
    return sorted(set(re.findall(r"\b(gkoc_\w+)\s*\(", pre)))


def test_header_is_plain_c():
    subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", HEADER], check=True)


def test_library_exports_every_declared_symbol():
    import ginkgo_amd as g
    assert os.path.exists(g.LIB_PATH), "run __graft_entry__.build() first"
    out = subprocess.run(["nm", "-D", "--defined-only", g.LIB_PATH],
                         capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (gkoc_\w+)", out))
    declared = declared_symbols()
    assert len(declared) > 100
    missing = [s for s in declared if s not in exported]
    assert not missing, f"declared but not exported: {missing}"
    undeclared = sorted(exported - set(declared))
    assert not undeclared, f"exported but not in the header: {undeclared}"


def test_library_loads_and_reports_version():
    from ginkgo_amd import _lib
    lib = _lib.lib()
    assert lib.gkoc_version() >= 1
    assert lib.gkoc_last_error() is not None


def test_no_torch_types_in_abi():
    src = open(HEADER).read()
    assert "torch" not in src and "at::" not in src and "std::" not in src


def test_executor_fails_loudly_without_gpu():
    import torch
    import ginkgo_amd as g
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(g.GkoError):
        g.Cdna4Executor.create(0)


def test_missing_library_is_an_error(monkeypatch):
    from ginkgo_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgko_cdna4.so")
    with pytest.raises(_lib.NotCompiled):
        _lib.lib()


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may
    touch oracle/"""
    pkg = os.path.join(ROOT, "ginkgo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.replace("INTEGRATION", ""), \
                    f"{os.path.join(dirpath, f)} mentions the oracle"


def test_counting_arguments_survive_the_call_tape():
    """host logic of ginkgo_amd._lib: the arguments of recorded calls are fixed, an argument that has
    to count (the exchange number of the one-kernel distributed product) is a ctypes integer that
    bump() increments - now and at the same point of every replay"""
    import ctypes as C
    from ginkgo_amd import _lib
    c = C.c_uint32(0)
    seen = []
    with _lib.record() as tape:
        _lib.bump(c)
        tape.calls.append((lambda v: seen.append(v.value) or 0, (c,), "observe"))   # a "call" taking c
        _lib.bump(c)
    assert c.value == 2 and [name for _, _, name in tape.calls] == ["bump", "observe", "bump"]
    tape.replay()
    tape.replay()
    assert c.value == 6 and seen == [3, 5]
    _lib.bump(c)                      # outside a recording: just counts
    assert c.value == 7


# namespaces of gko::kernels::hip that SURVEY.md 8 puts on the path (a1-a15) or next to it (f1-f4)
IN_SCOPE = ("csr", "ell", "sellp", "coo", "hybrid", "dense", "diagonal", "components", "cg", "fcg", "pipe_cg",
            "bicg", "bicgstab", "cgs", "gcr", "gmres", "common_gmres", "cb_gmres", "idr", "minres", "ir",
            "chebyshev", "jacobi", "residual_norm", "implicit_residual_norm", "set_all_statuses", "permutation",
            "scaled_permutation", "device_matrix_data", "distributed_matrix", "distributed_vector", "partition",
            "partition_helpers", "index_map", "assembly")
# what is knowingly left to Ginkgo's NotCompiled stubs there: conversions to the Fbcsr and
# SparsityCsr formats (out of scope), the lookup benchmark helper and the mixed-index form of
# convert_ptrs_to_idxs
STUBS_ALLOWED = {"csr::convert_to_fbcsr", "dense::convert_to_fbcsr", "dense::count_nonzero_blocks_per_row",
                 "dense::convert_to_sparsity_csr", "csr::benchmark_lookup", "components::convert_ptrs_to_idxs"}


def test_no_kernel_of_the_path_is_left_on_a_stub():
    """libginkgo_hip.so = this backend's strong definitions + Ginkgo's own stub object with every symbol
    weakened (gko_binding/build.py): a kernel template of an in-scope namespace that still shows up as a
    WEAK symbol - for any of float, double, complex<float>, complex<double>, int32, int64 - is a kernel
    this backend does not provide.  The list of those is short and fixed."""
    lib = os.path.join(ROOT, "ginkgo_amd", "lib", "libginkgo_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libginkgo_hip.so not built (needs the Ginkgo source tree: gko_binding/build.py)")
    out = subprocess.run(["nm", "-DC", lib], capture_output=True, text=True, check=True).stdout
    weak, strong = set(), set()
    for line in out.splitlines():
        m = re.search(r" ([WT]) (?:void |bool |int )?gko::kernels::hip::(\w+)::(\w+)[<(]", line)
        if m and m.group(2) in IN_SCOPE:
            (weak if m.group(1) == "W" else strong).add(f"{m.group(2)}::{m.group(3)}")
    # (weak-only helper templates of the binding itself - scratch getters and the like - are not kernels
    # of the reference: they have no declaration in core/**/_kernels.hpp and no strong twin anywhere)
    helpers = {"components::assembly_scratch", "components::complex_scratch", "coo::scratch",
               "jacobi::has_precisions"}
    stubs = weak - helpers
    assert stubs <= STUBS_ALLOWED, sorted(stubs - STUBS_ALLOWED)
    assert len(strong) >= 220, len(strong)
    # every value type: the path's kernels exist for the two complex types as for the real ones
    for kernel in ("csr::spmv", "ell::spmv", "sellp::spmv", "coo::spmv2", "cg::step_2", "jacobi::generate",
                   "idr::step_3", "minres::step_1", "csr::spgemm", "dense::apply", "cb_gmres::arnoldi"):
        pat = re.compile(r" T .*gko::kernels::hip::" + kernel.replace("::", "::") + r"<std::complex<double>")
        assert any(pat.search(l) for l in out.splitlines()), kernel + " has no complex<double> definition"
