"""Developer probe: host<->device copy rates for pageable vs registered host memory."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo
from cl_ops_amd._hip import lib
ctx = clo.Context(0); q = clo.Queue(ctx)
for mb in (16, 256, 1024):
    nbytes = mb << 20
    a = np.ones(nbytes, np.uint8); b = np.empty(nbytes, np.uint8)
    d = clo.Buffer(ctx, nbytes)
    def t(fn):
        t0 = time.perf_counter(); fn(); return (time.perf_counter() - t0) * 1e3
    def h2d(): lib.clo_hip_memcpy_h2d_async(d.ptr, a.ctypes.data, nbytes, q.stream); q.finish()
    def d2h(): lib.clo_hip_memcpy_d2h_async(b.ctypes.data, d.ptr, nbytes, q.stream); q.finish()
    h2d(); d2h()
    p_h2d = min(t(h2d) for _ in range(3)); p_d2h = min(t(d2h) for _ in range(3))
    reg = t(lambda: (lib.clo_hip_host_register(a.ctypes.data, nbytes), lib.clo_hip_host_register(b.ctypes.data, nbytes)))
    r_h2d = min(t(h2d) for _ in range(3)); r_d2h = min(t(d2h) for _ in range(3))
    # both directions at once on two streams
    q2 = clo.Queue(ctx)
    d2 = clo.Buffer(ctx, nbytes)
    def both():
        lib.clo_hip_memcpy_h2d_async(d.ptr, a.ctypes.data, nbytes, q.stream)
        lib.clo_hip_memcpy_d2h_async(b.ctypes.data, d2.ptr, nbytes, q2.stream)
        q.finish(); q2.finish()
    bo = min(t(both) for _ in range(3))
    unreg = t(lambda: (lib.clo_hip_host_unregister(a.ctypes.data), lib.clo_hip_host_unregister(b.ctypes.data)))
    gb = nbytes / 1e6
    print("%5d MiB: pageable h2d %.1f GB/s d2h %.1f GB/s | register(2 bufs) %.1f ms unregister %.1f ms | pinned h2d %.1f d2h %.1f GB/s, both at once %.1f ms (%.1f GB/s each)"
          % (mb, gb / p_h2d, gb / p_d2h, reg, unreg, gb / r_h2d, gb / r_d2h, bo, gb / bo), flush=True)
    d.close(); d2.close(); q2.close()
