"""Developer probe: do pageable async copies overlap (one thread / two threads)?"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo
from cl_ops_amd._hip import lib
ctx = clo.Context(0); q = clo.Queue(ctx); q2 = clo.Queue(ctx)
nbytes = 512 << 20
a = np.ones(nbytes, np.uint8); b = np.empty(nbytes, np.uint8)
d = clo.Buffer(ctx, nbytes); d2 = clo.Buffer(ctx, nbytes)
def t(fn):
    t0 = time.perf_counter(); fn(); return (time.perf_counter() - t0) * 1e3
def h2d(): lib.clo_hip_memcpy_h2d_async(d.ptr, a.ctypes.data, nbytes, q.stream)
def d2h(): lib.clo_hip_memcpy_d2h_async(b.ctypes.data, d2.ptr, nbytes, q2.stream)
def one_thread():
    h2d(); d2h(); q.finish(); q2.finish()
def two_threads():
    th = threading.Thread(target=lambda: (lib.clo_hip_set_device(0), d2h(), q2.finish()))
    th.start(); h2d(); q.finish(); th.join()
def chunked_one_thread(chunk=32 << 20):
    for off in range(0, nbytes, chunk):
        lib.clo_hip_memcpy_h2d_async(d.ptr + off, a.ctypes.data + off, chunk, q.stream)
        lib.clo_hip_memcpy_d2h_async(b.ctypes.data + off, d2.ptr + off, chunk, q2.stream)
    q.finish(); q2.finish()
one_thread()
print("issue time of one pageable h2d call: %.2f ms (copy itself ~%.1f ms)" % (t(h2d), nbytes / 56e6)); q.finish()
print("pageable, one thread, h2d+d2h: %.1f ms" % min(t(one_thread) for _ in range(3)))
print("pageable, two threads: %.1f ms" % min(t(two_threads) for _ in range(3)))
print("pageable, one thread, 32 MiB chunks interleaved: %.1f ms" % min(t(chunked_one_thread) for _ in range(3)))
lib.clo_hip_host_register(a.ctypes.data, nbytes); lib.clo_hip_host_register(b.ctypes.data, nbytes)
print("pinned, one thread: %.1f ms; issue time of one h2d call %.2f ms" % (min(t(one_thread) for _ in range(3)), t(h2d))); q.finish()
print("pinned, chunked interleaved: %.1f ms" % min(t(chunked_one_thread) for _ in range(3)))
