"""Developer probe: clo_sort_with_host_data / small clo_scan_with_host_data wall time, one vs two queues."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo
from cl_ops_amd.api import lib, vp, _Err
ctx = clo.Context(0); q = clo.Queue(ctx); q2 = clo.Queue(ctx)
for logn in (22, 26):
    n = 1 << logn
    a = np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    out = np.zeros(n, np.uint32)
    s = clo.Sorter("satradix", ctx, "uint")
    def run(comm):
        err = _Err()
        ok = lib.clo_sort_with_host_data(s.h, q.h, comm.h if comm else None, a.ctypes.data_as(vp), out.ctypes.data_as(vp), n, 0, err.ref)
        err.raise_if_set(); assert ok
    for comm in (None, q2):
        run(comm)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); run(comm); ts.append((time.perf_counter() - t0) * 1e3)
        print("2^%d uint host sort (%s): %.2f ms" % (logn, "one queue" if comm is None else "two queues", min(ts)), flush=True)
    assert np.all(out[:-1] <= out[1:])
    s.close()
    sc = clo.Scanner("blelloch", ctx, "uint", "uint")
    def runs(comm):
        err = _Err()
        ok = lib.clo_scan_with_host_data(sc.h, q.h, comm.h if comm else None, a.ctypes.data_as(vp), out.ctypes.data_as(vp), n, 0, err.ref)
        err.raise_if_set(); assert ok
    os.environ["CLO_SCAN_NO_PIPELINE"] = "1"
    for comm in (None, q2):
        runs(comm)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); runs(comm); ts.append((time.perf_counter() - t0) * 1e3)
        print("2^%d uint host scan, plain path (%s): %.2f ms" % (logn, "one queue" if comm is None else "two queues", min(ts)), flush=True)
    del os.environ["CLO_SCAN_NO_PIPELINE"]
    sc.close()
