"""ctypes loader for libgko_cdna4.so (the C ABI declared in include/gko_cdna4.h).

There is NO fallback: if the HIP library is missing or a call fails, a
GkoError is raised.  torch is imported first so that libamdhip64.so.7 /
librccl.so.1 resolve to the copies torch already mapped (one HIP runtime per
process).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below, see docstring)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libgko_cdna4.so")


class GkoError(RuntimeError):
    """Mirrors gko::Error (include/ginkgo/core/base/exception.hpp)."""


class NotCompiled(GkoError):
    """libgko_cdna4.so has not been built (gko::NotCompiled)."""


class NotSupported(GkoError):
    pass


class DimensionMismatch(GkoError):
    pass


class BadDimension(GkoError):
    """gko::BadDimension"""


class JacobiScheme(C.Structure):
    """gkoc_jacobi_scheme == gko block_interleaved_storage_scheme."""
    _fields_ = [("block_offset", C.c_int64), ("group_offset", C.c_int64),
                ("group_power", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NotCompiled(
                f"{LIB_PATH} not found - build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.gkoc_last_error.restype = C.c_char_p
        _lib.gkoc_reduction_workspace_bytes.restype = C.c_size_t
        _lib.gkoc_x_workspace_bytes.restype = C.c_size_t
        _lib.gkoc_coo_workspace_bytes.restype = C.c_size_t
    return _lib


def _conv(a):
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, torch.Tensor):
        return C.c_void_p(a.data_ptr())
    if isinstance(a, bool):
        return C.c_int64(int(a))
    if isinstance(a, int):
        return C.c_int64(a)
    return a  # already a ctypes object (c_double, c_float, Structure, byref)


def _raise(name, rc):
    msg = lib().gkoc_last_error().decode(errors="replace")
    if rc == -2:
        raise NotSupported(f"{name}: {msg}")
    raise GkoError(f"{name} failed with status {rc}: {msg}")


_tape = None


class Tape:
    """A recorded sequence of gkoc_* calls with their converted arguments, to be
    issued again without the Python work of building them (attribute look-ups,
    argument conversion: 8-15 us per call against ~1.5 us for the bare ctypes
    call).  Host-side analogue of a captured graph for loops that cannot be
    captured (collectives between the kernels, a criterion the host reads):
    valid while every buffer that was passed keeps its address - the tape holds
    a reference to each tensor argument so none is freed - and only for code
    whose device work goes through call() alone."""

    __slots__ = ("calls", "keep", "result")

    def __init__(self):
        self.calls, self.keep, self.result = [], [], None

    def replay(self):
        for fn, cargs, name in self.calls:
            rc = fn(*cargs)
            if rc != 0:
                _raise(name, rc)
        return self.result


class record:
    """with record() as tape: ...   every call() inside runs AND is recorded"""

    def __enter__(self):
        global _tape
        self.prev, self.tape = _tape, Tape()
        _tape = self.tape
        return self.tape

    def __exit__(self, *exc):
        global _tape
        _tape = self.prev
        return False


def call(name, *args):
    """Invoke a gkoc_* entry point; non-zero status raises."""
    fn = getattr(lib(), name)
    cargs = [_conv(a) for a in args]
    rc = fn(*cargs)
    if rc != 0:
        _raise(name, rc)
    if _tape is not None:
        _tape.calls.append((fn, cargs, name))
        _tape.keep.append(args)


def bump(counter):
    """counter.value += 1 (a ctypes integer that is passed by value to later calls), now and
    again at this point of every replay of the tape being recorded: arguments of recorded calls
    are fixed, an argument that must count is such an object"""
    def step():
        counter.value += 1
        return 0
    step()
    if _tape is not None:
        _tape.calls.append((step, (), "bump"))


VT = {torch.float64: "f64", torch.float32: "f32"}
IT = {torch.int32: "i32", torch.int64: "i64"}


def cval(dtype, v):
    return C.c_double(float(v)) if dtype == torch.float64 else C.c_float(float(v))
