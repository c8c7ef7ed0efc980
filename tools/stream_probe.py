"""Pair passes: one work-group per tile vs the streaming kernel (tickets, next tile requested
early). GPU box only. usage: python tools/stream_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLO_NO_WARMUP", "1")
os.environ["CLO_RADIX_SWEEP"] = "0"
import cl_ops_amd as clo  # noqa: E402
from cl_ops_amd import _hip  # noqa: E402
from cl_ops_amd._hip import lib  # noqa: E402

ctx = clo.Context(0)
q = clo.Queue(ctx)
rng = np.random.default_rng(0)
for kind, log2n in (("u32", 28), ("u64", 28), ("pairs", 28), ("u32", 24)):
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
    ref = None
    for mode in ("0", "1", "0", "1"):
        os.environ["CLO_RADIX_STREAM"] = mode
        s = clo.Sorter("satradix", ctx, et, **kw)
        for _ in range(3):
            s.with_device_data(q, bs, bd, n)
        q.finish()
        got = dst.cpu().numpy().view(host.dtype)
        if ref is None:
            ref = got.copy()
        ok = np.array_equal(got, ref)
        t = clo.HipEventTimer(q)
        t.start()
        for _ in range(20):
            s.with_device_data(q, bs, bd, n)
        t.stop()
        q.finish()
        ms = t.elapsed_ms() / 20
        lib.clo_hip_timing_reset()
        lib.clo_hip_timing_enable(1)
        for _ in range(5):
            s.with_device_data(q, bs, bd, n)
        q.finish()
        lib.clo_hip_timing_enable(0)
        c, tot = _hip.timing_read("radix_pass")
        print("%-5s 2^%d stream=%s: %.3f ms/sort %.0f Mkeys/s pass kernel %.4f ms same=%s" % (kind, log2n, mode, ms, n / ms / 1e3, tot / c, ok), flush=True)
        s.close()
    bs.close()
    bd.close()
    del src, dst
