"""Developer probe (GPU box): clo_hip_radix_sort vs clo_hip_radix_sort_fed (first digits handed over by the producer of the
keys) on resident data, back-to-back sorts. usage: python tools/fed_probe.py [log2n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo  # noqa: E402
from cl_ops_amd import _hip  # noqa: E402
from cl_ops_amd._hip import lib  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
ctx = clo.Context(0)
q = clo.Queue(ctx)
for kind, es, bits in (("uint32", 4, 32), ("uint64", 8, 64)):
    n = 1 << logn
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    t = torch.randint(-(1 << 31), 1 << 31, (n,), generator=g, device="cuda", dtype=torch.int32) if es == 4 else \
        torch.randint(-(1 << 62), 1 << 62, (n,), generator=g, device="cuda", dtype=torch.int64) * 2
    dig = (t & 0xff).to(torch.uint8)
    dst, tmp = torch.empty_like(t), torch.empty_like(t)
    wsb = lib.clo_hip_radix_workspace_bytes(n, es, bits, 4)
    ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    timer = clo.HipEventTimer(q)
    out = {}
    for name, d in (("plain", None), ("fed", dig.data_ptr()), ("plain", None), ("fed", dig.data_ptr())):
        for _ in range(3):
            _hip.check(lib.clo_hip_radix_sort_fed(t.data_ptr(), dst.data_ptr(), tmp.data_ptr(), n, es, 0, bits, 0, 4, d, ws.data_ptr(), wsb, q.stream))
        q.finish()
        timer.start()
        for _ in range(20):
            _hip.check(lib.clo_hip_radix_sort_fed(t.data_ptr(), dst.data_ptr(), tmp.data_ptr(), n, es, 0, bits, 0, 4, d, ws.data_ptr(), wsb, q.stream))
        timer.stop()
        q.finish()
        out.setdefault(name, []).append(timer.elapsed_ms() / 20)
    u = dst if es == 8 else (dst.to(torch.int64) & 0xFFFFFFFF)
    ok = bool((u[1:] >= u[:-1]).all()) if es == 4 else bool(((dst[1:] ^ (-(1 << 63))) >= (dst[:-1] ^ (-(1 << 63)))).all())
    print("2^%d %s: plain %s ms, fed %s ms -> %.0f / %.0f Mkeys/s; sorted=%s" % (logn, kind, ["%.3f" % x for x in out["plain"]], ["%.3f" % x for x in out["fed"]],
          n / min(out["plain"]) / 1e3, n / min(out["fed"]) / 1e3, ok), flush=True)
    del t, dst, tmp, ws, dig
