"""The floor of the mailbox all-reduce (csrc/comm_ipc.hpp): ONE process, a communicator of one rank - the
kernel stores its words into its own (uncached) window, polls them, sums: launch + store + poll + sum without
a peer.  What 8 GPUs add is one xGMI store latency (the peers' words arrive while this rank polls).
  python tools/mailbox_floor.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import ginkgo_amd as g
from ginkgo_amd._lib import call

ex = g.Cdna4Executor.create(0)
h = C.c_void_p(0)
mine = (C.c_uint8 * 128)()
call("gkoc_comm_ipc_create", C.byref(h), C.c_int(1), C.c_int(0), C.c_int64(1 << 20), mine)
call("gkoc_comm_ipc_connect", h, mine)
for n in (2, 3, 32):
    t = torch.ones(n, dtype=torch.float64, device=ex.device)
    for _ in range(200):
        call("gkoc_comm_all_reduce_sum", h, ex.stream, t, n, C.c_size_t(8))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5000
    e0.record()
    for _ in range(reps):
        call("gkoc_comm_all_reduce_sum", h, ex.stream, t, n, C.c_size_t(8))
    e1.record()
    torch.cuda.synchronize()
    print(f"mailbox all-reduce of {n} doubles, one rank, back to back on one stream: "
          f"{e0.elapsed_time(e1) / reps * 1e3:.2f} us per call (device clock)")
st = C.c_uint32(0)
call("gkoc_comm_status", h, C.byref(st))
print("status", st.value)
call("gkoc_comm_destroy", h)
