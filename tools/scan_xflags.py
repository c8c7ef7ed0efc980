"""2^26 / 2^28 uint32 scans with the experiment flags of clo_hip_scan.hip (CLO_SCAN_XFLAGS:
2 = non-temporal loads, 4 = non-temporal stores). GPU box only."""
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
for log2n in (26, 28):
    n = 1 << log2n
    a = np.random.default_rng(0).integers(0, 128, n, dtype=np.uint32)
    src = torch.from_numpy(a.view(np.int32)).cuda()
    dst = torch.empty_like(src)
    bs, bd = clo.Buffer(ctx, n * 4, device_ptr=src.data_ptr()), clo.Buffer(ctx, n * 4, device_ptr=dst.data_ptr())
    sc = clo.Scanner("blelloch", ctx, "uint", "uint")
    exp = (np.cumsum(a, dtype=np.uint64) - a).astype(np.uint32)
    for flags in ("0", "2", "4", "6", "0"):
        os.environ["CLO_SCAN_XFLAGS"] = flags
        for _ in range(5):
            sc.with_device_data(q, bs, bd, n)
        q.finish()
        t = clo.HipEventTimer(q)
        t.start()
        for _ in range(100):
            sc.with_device_data(q, bs, bd, n)
        t.stop()
        q.finish()
        ms = t.elapsed_ms() / 100
        ok = np.array_equal(dst.cpu().numpy().view(np.uint32), exp)
        print("2^%d xflags=%s: %.4f ms/scan  %.0f MValues/s  %.2f TB/s  ok=%s" % (log2n, flags, ms, n / ms / 1e3, 8 * n / ms / 1e9, ok), flush=True)
    sc.close()
