"""Developer probe (GPU box): satradix with every value of its `radix` option, device-resident, back-to-back.
usage: python tools/radix_option_probe.py [log2n=26] [uint|ulong]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
et = sys.argv[2] if len(sys.argv) > 2 else "uint"
dt = clo.api.CLO_TYPE_NP[et]
n = 1 << logn
ctx = clo.Context(0)
q = clo.Queue(ctx)
t = clo.HipEventTimer(q)
a = np.random.default_rng(0).integers(0, int(np.iinfo(dt).max), n, dtype=np.uint64, endpoint=True).astype(dt)
exp = np.sort(a)
src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
src.write(q, a)
for radix in (2, 4, 8, 16, 32, 64, 128, 256):
    s = clo.Sorter("satradix", ctx, et, options="radix=%d" % radix)
    for _ in range(2):
        s.with_device_data(q, src, dst, n)
    q.finish()
    ms = []
    for _ in range(5):
        t.start()
        s.with_device_data(q, src, dst, n)
        t.stop()
        ms.append(t.elapsed_ms())
    ok = np.array_equal(dst.read(q, dt, n), exp)
    print("2^%d %s radix=%3d: %.3f ms -> %7.0f Mkeys/s  %s" % (logn, et, radix, min(ms), n / min(ms) / 1e3, "" if ok else "WRONG"), flush=True)
    s.close()
