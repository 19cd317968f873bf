"""Executor mirror: the MI355X counterpart of gko::HipExecutor.

Reference interface: include/ginkgo/core/base/executor.hpp:1785-1990
(HipExecutor::create / synchronize / get_num_devices / get_stream, exec_info
fields num_computing_units / max_subgroup_size).  Device memory and streams
are torch's (plumbing); all numerical work goes through libgko_cdna4.so.
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
    def alloc(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def to_device(self, array):
        if isinstance(array, torch.Tensor):
            return array.to(self.device)
        return torch.from_numpy(np.ascontiguousarray(array)).to(self.device)
