"""Read-only streaming rate of this GPU through the library's own reduce kernel
(one read of the array, nothing written): the ceiling for the per-tile histogram kernel.
GPU box only. usage: python tools/read_bw_probe.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cl_ops_amd import _hip  # noqa: E402

lib = _hip.lib
lib.clo_hip_reduce_sum.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
for log2n in (26, 28, 29):
    n = 1 << log2n
    x = torch.randint(0, 1 << 31, (n,), device="cuda", dtype=torch.int32)
    tot = torch.zeros(1, device="cuda", dtype=torch.int64)
    for _ in range(3):
        lib.clo_hip_reduce_sum(x.data_ptr(), n, 4, 0, tot.data_ptr(), None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        lib.clo_hip_reduce_sum(x.data_ptr(), n, 4, 0, tot.data_ptr(), torch.cuda.current_stream().cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("reduce of 2^%d uint32: %.4f ms per call (memset + kernel), %.0f GB/s read" % (log2n, ms, n * 4 / ms / 1e6), flush=True)
    # torch's own copy as a second yardstick (read + write)
    y = torch.empty_like(x)
    y.copy_(x)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("  torch copy of the same array: %.4f ms, %.0f GB/s read + write" % (ms, 2 * n * 4 / ms / 1e6), flush=True)
    del x, y
