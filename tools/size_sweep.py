"""Developer probe: device time across sizes (device data, back-to-back calls), every power of two and the sizes just
around the switches between code paths — is anything slower than a LARGER array? GPU box only.
usage: python tools/size_sweep.py [satradix|abitonic|scan] [uint|ulong] [lo=10] [hi=25]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo  # noqa: E402

alg = sys.argv[1] if len(sys.argv) > 1 else "satradix"
et = sys.argv[2] if len(sys.argv) > 2 else "uint"
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 10
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 25
dt = clo.api.CLO_TYPE_NP[et]
ctx = clo.Context(0)
q = clo.Queue(ctx)
t = clo.HipEventTimer(q)
sizes = []
for logn in range(lo, hi + 1):
    n = 1 << logn
    sizes += [n] if alg == "abitonic" else [n - n // 8, n, n + 1]
obj = clo.Scanner("blelloch", ctx, et, "ulong" if et == "ulong" else "uint") if alg == "scan" else clo.Sorter(alg, ctx, et)
prev = None
rng = np.random.default_rng(0)
for n in sizes:
    a = (rng.integers(0, 128, n) if alg == "scan" else rng.integers(0, int(np.iinfo(dt).max), n, dtype=np.uint64, endpoint=True)).astype(dt)
    src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, max(a.nbytes, n * 8))
    src.write(q, a)
    for _ in range(3):
        obj.with_device_data(q, src, dst, n)
    q.finish()
    ms = []
    for _ in range(7):
        t.start()
        obj.with_device_data(q, src, dst, n)
        t.stop()
        ms.append(t.elapsed_ms())
    best = min(ms)
    flag = "   <-- a smaller array takes LONGER: %.4f ms at n = %d" % (prev[1], prev[0]) if prev is not None and prev[1] > best * 1.07 else ""
    print("n = %9d: %.4f ms -> %8.0f M/s%s" % (n, best, n / best / 1e3, flag), flush=True)
    prev = (n, best)
    src.close()
    dst.close()
