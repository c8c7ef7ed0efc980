import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import cl_ops_amd as clo
ctx = clo.Context(0); q = clo.Queue(ctx)
n = 1 << 26
rng = np.random.default_rng(0)
keys = rng.integers(0, 2**32, n, dtype=np.uint64)
a = (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)
for alg in ("abitonic", "satradix"):
    s = clo.Sorter(alg, ctx, "ulong", key_type="uint", get_key="(uint) ((x) >> 32)")
    src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
    src.write(q, a)
    s.with_device_data(q, src, dst, n); q.finish()
    t = clo.HipEventTimer(q); ms = []
    for _ in range(3):
        t.start(); s.with_device_data(q, src, dst, n); t.stop(); ms.append(t.elapsed_ms())
    t.close()
    print("%s pairs 2^26: %.3f ms -> %.0f Mkeys/s" % (alg, min(ms), n / min(ms) / 1e3), flush=True)
    src.close(); dst.close(); s.close()
from cl_ops_amd import _hip
lib = _hip.lib
s = clo.Sorter("abitonic", ctx, "ulong", key_type="uint", get_key="(uint) ((x) >> 32)")
src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
src.write(q, a)
s.with_device_data(q, src, dst, n); q.finish()
lib.clo_hip_timing_enable(1); lib.clo_hip_timing_reset()
s.with_device_data(q, src, dst, n); q.finish()
print("abitonic pairs kernels:", {l: _hip.timing_read(l) for l in ("bitonic_presort", "bitonic_tile", "bitonic_strided")}, flush=True)
lib.clo_hip_timing_enable(0)
