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
