"""Developer probe: where does the pipelined host-data sort differ from numpy?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo
ctx = clo.Context(0)
qx, qc = clo.Queue(ctx, profiling=True), clo.Queue(ctx)
n = 1 << 24
a = np.random.default_rng(0).integers(0, 1 << 32, n, dtype=np.uint32)
s = clo.Sorter("satradix", ctx, "uint")
for queues in ((qx, qc), (qx, None)):
    got = s.with_host_data(a, *queues)
    ref = np.sort(a)
    bad = np.flatnonzero(got != ref)
    print("queues", "two" if queues[1] else "one", "mismatches:", bad.size, "first:", bad[:5], "last:", bad[-5:] if bad.size else "")
    if bad.size:
        b = np.searchsorted(ref >> 28, np.arange(17))
        print("bucket starts", b)
        print("sorted within buckets?", [bool(np.all(np.diff(got[b[i]:b[i + 1]].astype(np.int64)) >= 0)) for i in range(16)])
        print("top nibble ok?", [bool(np.all((got[b[i]:b[i + 1]] >> 28) == i)) for i in range(16)])
# in place on the host (the harness): same array in and out
x = a.copy()
from cl_ops_amd.api import lib, vp, _Err
err = _Err()
ok = lib.clo_sort_with_host_data(s.h, qx.h, qc.h, x.ctypes.data_as(vp), x.ctypes.data_as(vp), n, 0, err.ref)
err.raise_if_set()
print("in place on the host:", bool(np.array_equal(x, np.sort(a))))
