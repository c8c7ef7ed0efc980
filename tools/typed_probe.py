"""Developer probe: abitonic / satradix on typed keys, device-resident, 2^26 elements."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo
ctx = clo.Context(0); q = clo.Queue(ctx)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << logn
rng = np.random.default_rng(0)
for alg in ("abitonic", "satradix"):
    for et in ("uint", "int", "float", "ulong", "double"):
        dt = clo.api.CLO_TYPE_NP[et]
        if np.issubdtype(dt, np.floating):
            a = ((rng.random(n) - 0.5) * 1e6).astype(dt)
        else:
            a = rng.integers(0, 2**31, n, dtype=np.int64).astype(dt)
        s = clo.Sorter(alg, ctx, et)
        src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
        src.write(q, a)
        s.with_device_data(q, src, dst, n); q.finish()
        t = clo.HipEventTimer(q); ms = []
        for _ in range(3):
            t.start(); s.with_device_data(q, src, dst, n); t.stop(); ms.append(t.elapsed_ms())
        t.close()
        got = dst.read(q, dt, n)
        print("%s %s 2^%d: %.3f ms -> %.0f Mkeys/s sorted=%s" % (alg, et, logn, min(ms), n / min(ms) / 1e3, bool(np.all(got[:-1] <= got[1:]))), flush=True)
        src.close(); dst.close(); s.close()

# per-kernel time of abitonic on 8-byte keys
from cl_ops_amd import _hip
lib = _hip.lib
a = rng.integers(0, 2**62, n, dtype=np.int64).astype(np.uint64)
s = clo.Sorter("abitonic", ctx, "ulong")
src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
src.write(q, a)
s.with_device_data(q, src, dst, n); q.finish()
lib.clo_hip_timing_enable(1); lib.clo_hip_timing_reset()
s.with_device_data(q, src, dst, n); q.finish()
print("abitonic ulong kernels:", {l: _hip.timing_read(l) for l in ("bitonic_presort", "bitonic_tile", "bitonic_strided")}, flush=True)
lib.clo_hip_timing_enable(0)
