import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, ctypes as C
import cl_ops_amd as clo
from cl_ops_amd import _hip
lib = _hip.lib
ctx = clo.Context(0); q = clo.Queue(ctx)
n = 1 << 28
a = np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
src = clo.Buffer(ctx, a.nbytes); src.write(q, a)
tot = clo.Buffer(ctx, 8)
lib.clo_hip_timing_enable(1)
for it in range(8):
    if it == 2:
        q.finish(); lib.clo_hip_timing_reset()     # two warm-up calls
    tot.write(q, np.zeros(1, np.uint64))
    st = lib.clo_hip_reduce_sum(C.c_void_p(src.ptr), C.c_size_t(n), 4, 0, C.c_void_p(tot.ptr), C.c_void_p(q.stream))
    assert st == 0, st
q.finish()
c, t = _hip.timing_read("reduce")
print("reduce 2^28 u32: %.4f ms avg over %d -> %.2f TB/s" % (t / c, c, n * 4 / (t / c * 1e-3) / 1e12))
got = tot.read(q, np.uint64, 1)[0]
print("ok", int(got) == int(a.sum(dtype=np.uint64)))
