"""Developer helper for counter runs: three satradix sorts of 2^LOGN uint keys."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << logn
ctx = clo.Context(0); q = clo.Queue(ctx)
a = np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
s = clo.Sorter("satradix", ctx, "uint")
src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
src.write(q, a)
for _ in range(3):
    s.with_device_data(q, src, dst, n)
q.finish()
print("done")
