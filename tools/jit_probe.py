"""Developer probe (GPU box): abitonic with identity keys (ahead-of-time min / max networks), a partial key (ahead-of-time
general compare) and two expressions that go through hiprtc. usage: python tools/jit_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CLO_NO_WARMUP"] = "1"
import cl_ops_amd as clo
ctx = clo.Context(0); q = clo.Queue(ctx)
for logn in ((26,) if os.environ.get("JIT_PROBE_ONLY26") else (20, 24, 26)):
    n = 1 << logn
    a = np.random.default_rng(0).integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    for name, kw in (("identity (AOT min/max)", {}), ("partial key (AOT general compare)", dict(get_key="((x) >> 8)")), ("hiprtc ((x) ^ 0x55)", dict(get_key="((x) ^ 0x55)")),
                     ("hiprtc compare a%7 (partial order)", dict(compare="(((a) >> 3) > ((b) >> 3))"))):
        t0 = time.perf_counter()
        s = clo.Sorter("abitonic", ctx, "uint", **kw)
        t_new = time.perf_counter() - t0
        buf = clo.Buffer(ctx, n * 4)
        ts = []
        for rep in range(4):
            buf.write(q, a)
            q.finish()
            t0 = time.perf_counter()
            s.with_device_data(q, buf, None, n)
            q.finish()
            ts.append((time.perf_counter() - t0) * 1e3)
        got = buf.read(q, np.uint32, n)
        key = got if "get_key" not in kw and "compare" not in kw else (got >> np.uint32(8) if kw.get("get_key") == "((x) >> 8)" else (got ^ np.uint32(0x55) if "get_key" in kw else got >> np.uint32(3)))
        print("2^%d %-40s %.3f ms (%.0f Mkeys/s); clo_sort_new %.2f s; ordered=%s" % (logn, name, min(ts), n / min(ts) / 1e3, t_new, bool(np.all(key[:-1] <= key[1:]))), flush=True)
        buf.close(); s.close()
