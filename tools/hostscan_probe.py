"""Developer probe: clo_scan_with_host_data wall time into a pre-touched output
(pipelined above 2^24 elements; CLO_SCAN_NO_PIPELINE=1 forces the plain path)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo
from cl_ops_amd.api import lib, vp, _Err
ctx = clo.Context(0); q = clo.Queue(ctx); q2 = clo.Queue(ctx)
for logn in (23, 24, 26, 28):
    n = 1 << logn
    a = np.random.default_rng(0).integers(0, 128, n, dtype=np.uint32)
    out = np.zeros(n, np.uint32)
    sc = clo.Scanner("blelloch", ctx, "uint", "uint")
    def run(comm):
        err = _Err()
        ok = lib.clo_scan_with_host_data(sc.h, q.h, comm.h if comm else None, a.ctypes.data_as(vp), out.ctypes.data_as(vp), n, 0, err.ref)
        err.raise_if_set(); assert ok
    for comm in (None, q2):
        run(comm)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); run(comm); ts.append((time.perf_counter() - t0) * 1e3)
        print("2^%d uint->uint host scan (%s): %.2f ms (%.1f GB/s of in+out)" % (logn, "one queue" if comm is None else "two queues", min(ts), 8 * n / min(ts) / 1e6), flush=True)
    exp_last = int(a[:-1].sum(dtype=np.uint64)) & 0xFFFFFFFF
    assert int(out[-1]) == exp_last
    sc.close()
