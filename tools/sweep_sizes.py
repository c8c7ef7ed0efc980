"""ms per sort, chain-free pair passes vs single-sweep passes, over sizes and element kinds
(device-resident, back-to-back sorts). GPU box only. usage: python tools/sweep_sizes.py [big]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLO_NO_WARMUP", "1")
import cl_ops_amd as clo  # noqa: E402

ctx = clo.Context(0)
q = clo.Queue(ctx)
rng = np.random.default_rng(0)


def bench(kind, log2n):
    n = 1 << log2n
    if kind == "u32":
        host, et, kw = rng.integers(0, 1 << 32, n, dtype=np.uint32), "uint", {}
    elif kind == "u64":
        host, et, kw = rng.integers(0, np.iinfo(np.uint64).max, n, dtype=np.uint64, endpoint=True), "ulong", {}
    else:
        host = (rng.integers(0, 1 << 32, n, dtype=np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        et, kw = "ulong", dict(key_type="uint", get_key="(uint) ((x) >> 32)")
    es = host.dtype.itemsize
    src = torch.from_numpy(host.view(np.int32 if es == 4 else np.int64)).cuda()
    dst = torch.empty_like(src)
    bs, bd = clo.Buffer(ctx, n * es, device_ptr=src.data_ptr()), clo.Buffer(ctx, n * es, device_ptr=dst.data_ptr())
    out = []
    for mode in ("0", "1"):
        os.environ["CLO_RADIX_SWEEP"] = mode
        s = clo.Sorter("satradix", ctx, et, **kw)
        for _ in range(3):
            s.with_device_data(q, bs, bd, n)
        q.finish()
        steps = max(5, min(200, (1 << 29) // n))
        t = clo.HipEventTimer(q)
        t.start()
        for _ in range(steps):
            s.with_device_data(q, bs, bd, n)
        t.stop()
        q.finish()
        out.append(t.elapsed_ms() / steps)
        s.close()
    print("%-6s 2^%-2d  pair %9.4f ms  sweep %9.4f ms  sweep/pair %.3f" % (kind, log2n, out[0], out[1], out[1] / out[0]), flush=True)
    bs.close()
    bd.close()


if len(sys.argv) > 1 and sys.argv[1] == "edge":   # around the switch between the two paths
    for rep in range(2):
        for kind, ls in (("u32", (17, 18, 19, 20, 21, 22, 23)), ("u64", (17, 18, 19, 20, 21, 22, 23)), ("pairs", (18, 20, 21, 22, 23))):
            for l in ls:
                bench(kind, l)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "big":   # where should the default switch back for 8-byte elements?
    for rep in range(2):
        for kind in ("u64", "pairs", "u32"):
            for l in (22, 23, 24, 25, 26, 27, 28):
                bench(kind, l)
    sys.exit(0)
for l in (14, 15, 16, 17, 18, 19, 20, 22, 24, 26, 28):
    bench("u32", l)
for l in (20, 24, 28):
    bench("u64", l)
for l in (24, 28):
    bench("pairs", l)
