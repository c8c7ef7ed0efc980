"""Developer probe: clo_sort_with_host_data host-to-host wall time, pipelined (chunks split as they arrive,
buckets sorted and copied out one after the other) vs upstream's blocking path (the default; the pipeline: CLO_SORT_HOST_PIPELINE=1), next to
the bare copies. GPU box only. usage: python tools/hostsort_pipe_probe.py [log2n ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo  # noqa: E402
from cl_ops_amd.api import lib, vp, _Err  # noqa: E402

ctx = clo.Context(0)
qx, qc = clo.Queue(ctx), clo.Queue(ctx)
for logn in [int(x) for x in sys.argv[1:]] or [24, 26, 28]:
    for kind in ("uint", "ulong"):
        n = 1 << logn
        dt = np.uint32 if kind == "uint" else np.uint64
        a = np.random.default_rng(0).integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
        buf = clo.Buffer(ctx, a.nbytes)

        def best(fn, reps=4):
            fn()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                ts.append((time.perf_counter() - t0) * 1e3)
            return min(ts)

        t_in = best(lambda: buf.write(qc, a))
        tmp = np.empty_like(a)
        tmp[:] = 0

        def rd():
            err = _Err()
            lib.ccl_buffer_enqueue_read(buf.h, qc.h, 1, 0, a.nbytes, tmp.ctypes.data_as(vp), None, err.ref)
            err.raise_if_set()
        t_out = best(rd)
        out = np.empty_like(a)
        out[:] = 0                                                   # (touched: no first-touch page faults inside the timings)

        def run():
            err = _Err()
            ok = lib.clo_sort_with_host_data(s.h, qx.h, qc.h, a.ctypes.data_as(vp), out.ctypes.data_as(vp), n, 0, err.ref)
            err.raise_if_set()
            assert ok
        os.environ["CLO_SORT_HOST_PIPELINE"] = "0"           # (read when the sorter is made)
        s = clo.Sorter("satradix", ctx, kind)
        t_block = best(run)
        s.close()
        os.environ["CLO_SORT_HOST_PIPELINE"] = "1"
        s = clo.Sorter("satradix", ctx, kind)
        t_pipe = best(run)
        del os.environ["CLO_SORT_HOST_PIPELINE"]
        assert np.array_equal(out[:1000], np.sort(a)[:1000])
        print("2^%d %-5s copy in %.2f ms + copy out %.2f ms = %.2f; pipelined %.2f ms (+%.2f); blocking %.2f ms (+%.2f)"
              % (logn, kind, t_in, t_out, t_in + t_out, t_pipe, t_pipe - t_in - t_out, t_block, t_block - t_in - t_out), flush=True)
        buf.close()
        s.close()
