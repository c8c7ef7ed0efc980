"""Developer A/B: satradix radix=16, single-digit passes (CLO_RADIX_PAIRS=0) vs digit pairs, size sweep."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo
ctx = clo.Context(0); q = clo.Queue(ctx)
t = clo.HipEventTimer(q)
for et, dt, kw in (("uint", np.uint32, {}), ("ulong", np.uint64, {}), ("ulong", np.uint64, dict(key_type="uint", get_key="(uint) ((x) >> 32)"))):
    for logn in (13, 16, 20, 24, 26, 28):
        n = 1 << logn
        a = np.random.default_rng(0).integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
        s = clo.Sorter("satradix", ctx, et, **kw)
        src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
        src.write(q, a)
        res = {}
        for mode in ("0", "1"):
            os.environ["CLO_RADIX_PAIRS"] = mode
            for _ in range(3):
                s.with_device_data(q, src, dst, n)
            q.finish()
            ms = []
            for _ in range(5):
                t.start(); s.with_device_data(q, src, dst, n); t.stop(); ms.append(t.elapsed_ms())
            res[mode] = min(ms)
            got = dst.read(q, dt, n)
            assert np.all((got[:-1] >> (32 if kw else 0)) <= (got[1:] >> (32 if kw else 0)))
        print("%s%s 2^%d: single %.4f ms, pairs %.4f ms (%+.1f %%)" % (et, "(pairs kv)" if kw else "", logn, res["0"], res["1"], 100 * (res["0"] / res["1"] - 1)), flush=True)
        src.close(); dst.close(); s.close()
