"""Developer diagnostic: phase stamps (s_memtime) of the radix pass kernel."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo
from cl_ops_amd._hip import lib

ctx = clo.Context(0); q = clo.Queue(ctx)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << logn
a = np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
s = clo.Sorter("satradix", ctx, "uint")
src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
src.write(q, a)
s.with_device_data(q, src, dst, n); q.finish()
dbg = clo.Buffer(ctx, 32768 * 8 * 8)
lib.clo_hip_radix_set_debug_buffer(dbg.ptr)
s.with_device_data(q, src, dst, n); q.finish()
lib.clo_hip_radix_set_debug_buffer(None)
st = dbg.read(q, np.uint64, 32768 * 8).reshape(-1, 8).astype(np.int64)   # stamps of the LAST pass
ntiles = min(32768, (n + 4095) // 4096)
st = st[:ntiles]
d = np.diff(st, axis=1)
names = ["load", "count+wave scan", "barrier", "wave bases+barrier", "scatter to LDS", "barrier", "read-out+stores"]
print("tiles", ntiles, "clock ticks (100MHz s_memtime? see below)")
for k, nm in enumerate(names):
    print("%-20s median %8.0f  p10 %8.0f  p90 %8.0f" % (nm, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
tot = st[:, 7] - st[:, 0]
print("total per tile median %.0f p90 %.0f ; kernel span %.0f" % (np.median(tot), np.percentile(tot, 90), st[:, 7].max() - st[:, 0].min()))
