"""Developer probe: satradix device time across sizes (device data, back-to-back)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo
ctx = clo.Context(0); q = clo.Queue(ctx)
t = clo.HipEventTimer(q)
for logn in (12, 13, 14, 16, 18, 20, 22, 24):
    n = 1 << logn
    a = np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    s = clo.Sorter("satradix", ctx, "uint")
    src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
    src.write(q, a)
    for _ in range(3):
        s.with_device_data(q, src, dst, n)
    q.finish()
    ms = []
    for _ in range(7):
        t.start(); s.with_device_data(q, src, dst, n); t.stop(); ms.append(t.elapsed_ms())
    assert np.array_equal(dst.read(q, np.uint32, n), np.sort(a))
    print("2^%d: %.4f ms -> %.0f Mkeys/s" % (logn, min(ms), n / min(ms) / 1e3), flush=True)
    src.close(); dst.close(); s.close()
