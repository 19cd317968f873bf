"""Executor mirror: the MI355X counterpart of gko::HipExecutor.

Reference interface: include/ginkgo/core/base/executor.hpp:1785-1990
(HipExecutor::create / synchronize / get_num_devices / get_stream, exec_info
fields num_computing_units / max_subgroup_size).  Streams are torch's
(plumbing); device memory of 1 MiB and more comes from the library's arena
(`gkoc_malloc_role`, csrc/arena.hip - the counterpart of HipExecutor::raw_alloc)
and is handed to torch as a tensor over that memory, so that matrix values, index
arrays and vectors live in different memory classes of the MI355X;
all numerical work goes through libgko_cdna4.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


class DeviceInfo(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("num_cu", C.c_int32),
                ("wave_size", C.c_int32), ("num_xcd", C.c_int32),
                ("max_threads_per_block", C.c_int32), ("major", C.c_int32),
                ("minor", C.c_int32), ("lds_bytes_per_cu", C.c_int32),
                ("hbm_bytes", C.c_int64), ("arch", C.c_char * 64)]


# roles of gkoc_malloc_role (include/gko_cdna4.h)
MEM_AUTO, MEM_VALUES, MEM_INDICES, MEM_VECTOR = 0, 1, 2, 3
_ARENA_MIN_BYTES = 1 << 20
_TYPESTR = {torch.float64: "<f8", torch.float32: "<f4", torch.float16: "<f2",
            torch.int64: "<i8", torch.int32: "<i4", torch.int16: "<i2", torch.int8: "|i1",
            torch.uint8: "|u1", torch.bool: "|b1"}


class _ArenaBlock:
    """Owner of one gkoc_malloc_role allocation; torch keeps it alive through the
    CUDA array interface and gkoc_free runs when the last tensor over it is gone."""

    def __init__(self, nbytes, role, shape, typestr):
        p = C.c_void_p()
        _lib.call("gkoc_malloc_role", C.byref(p), C.c_size_t(nbytes), C.c_int(role))
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (self.ptr, False), "version": 2}

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr:
            try:
                _lib.lib().gkoc_free(C.c_void_p(ptr))
            except Exception:   # interpreter shutdown
                pass


class Cdna4Executor:
    """`Cdna4Executor.create(device_id)` ~ `gko::HipExecutor::create(id, master)`."""

    def __init__(self, device_id=0):
        _lib.lib()  # fail loudly when the HIP library is missing
        n = C.c_int(0)
        _lib.call("gkoc_get_num_devices", C.byref(n))
        if n.value == 0 or not torch.cuda.is_available():
            raise _lib.GkoError(
                "Cdna4Executor: no HIP device visible (gko::HipError); this "
                "backend has no CPU fallback")
        if not 0 <= device_id < n.value:
            raise _lib.GkoError(f"invalid device id {device_id} (have {n.value})")
        self.device_id = device_id
        self.device = torch.device("cuda", device_id)
        info = DeviceInfo()
        _lib.call("gkoc_get_device_info", C.c_int(device_id), C.byref(info))
        self.info = info

    @staticmethod
    def create(device_id=0):
        return Cdna4Executor(device_id)

    @staticmethod
    def get_num_devices():
        n = C.c_int(0)
        _lib.call("gkoc_get_num_devices", C.byref(n))
        return n.value

    # --- exec_info accessors (executor.hpp:907-1030)
    def get_num_multiprocessor(self):
        return self.info.num_cu

    def get_warp_size(self):
        return self.info.wave_size

    def get_description(self):
        return f"Cdna4Executor on device {self.device_id} ({self.info.arch.decode()})"

    @property
    def stream(self):
        """hipStream_t (as int) all kernels are enqueued on."""
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self):
        _lib.call("gkoc_stream_synchronize", self.stream)

    # --- memory
    _arena_ok = True

    def alloc(self, shape, dtype, role=MEM_VECTOR):
        """uninitialised device array; `role` places it (MEM_VALUES / MEM_INDICES for
        the big read-only arrays of a matrix, MEM_VECTOR for everything kernels write)"""
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        n = 1
        for v in shape:
            n *= v
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        if nbytes < _ARENA_MIN_BYTES or dtype not in _TYPESTR or not Cdna4Executor._arena_ok:
            return torch.empty(shape, dtype=dtype, device=self.device)
        with torch.cuda.device(self.device):
            block = _ArenaBlock(nbytes, role, shape, _TYPESTR[dtype])
            try:
                t = torch.as_tensor(block, device=self.device)
            except Exception as e:   # torch cannot wrap foreign device memory here
                import warnings
                warnings.warn(f"gko-cdna4: arena memory not usable from torch ({e}); "
                              "falling back to torch's allocator")
                Cdna4Executor._arena_ok = False
                return torch.empty(shape, dtype=dtype, device=self.device)
        assert t.data_ptr() == block.ptr and t.dtype == dtype
        return t

    def zeros(self, shape, dtype, role=MEM_VECTOR):
        return self.alloc(shape, dtype, role).zero_()

    def to_device(self, array, role=MEM_VECTOR):
        if isinstance(array, torch.Tensor):
            src = array
        else:
            src = torch.from_numpy(np.ascontiguousarray(array))
        if src.device == self.device and src.is_contiguous():
            return src
        out = self.alloc(tuple(src.shape), src.dtype, role)
        out.copy_(src)
        return out

    def arena_info(self):
        """gkoc_arena_stats as a dict (include/gko_cdna4.h gkoc_arena_info)"""
        class Info(C.Structure):
            _fields_ = [("mode", C.c_int32), ("num_classes", C.c_int32),
                        ("chunk_bytes", C.c_int64), ("num_chunks", C.c_int64),
                        ("reserved_bytes", C.c_int64), ("used_bytes", C.c_int64),
                        ("num_allocations", C.c_int64), ("probes", C.c_int64),
                        ("granules_walked", C.c_int64), ("spare_bytes", C.c_int64),
                        ("class_reserved_bytes", C.c_int64 * 3),
                        ("class_used_bytes", C.c_int64 * 3),
                        ("granules_classified", C.c_int64), ("search_ns", C.c_int64),
                        ("probe_retries", C.c_int64), ("surveyed", C.c_int64),
                        ("search_budget_ms", C.c_int64), ("search_budget_spent", C.c_int64),
                        ("granules_unclassified", C.c_int64)]
        info = Info()
        with torch.cuda.device(self.device):
            _lib.call("gkoc_arena_stats", C.byref(info))
        out = {name: getattr(info, name) for name, _ in Info._fields_}
        out["class_reserved_bytes"] = list(info.class_reserved_bytes)
        out["class_used_bytes"] = list(info.class_used_bytes)
        return out

    def memory_class(self, tensor):
        """memory class (0..2) of a tensor inside the arena's class regions, else -1"""
        c = C.c_int(-1)
        _lib.call("gkoc_arena_class_of", C.c_void_p(tensor.data_ptr()), C.byref(c))
        return c.value
