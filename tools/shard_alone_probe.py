"""Developer probe (GPU box): the sharded sort's whole protocol on ONE rank that sends its keys to itself over RCCL
(`loopback=1`) — partition on 8 bits, count all-gather, sliced grouped send/recv on the transfer stream, one segmented
sort per slice on the exec stream — against the plain local sort. What a node adds is xGMI instead of a device copy.
usage: python tools/shard_alone_probe.py [log2n=28] [uint|ulong|both]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLO_NO_WARMUP", "1")
from cl_ops_amd.multigpu import CShardedSorter  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
which = sys.argv[2] if len(sys.argv) > 2 else "both"
n = 1 << logn
for etype in (("uint", "ulong") if which == "both" else (which,)):
    if etype == "uint":
        a = np.random.default_rng(0).integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        t = torch.from_numpy(a.view(np.int32)).cuda()
    else:
        a = np.random.default_rng(0).integers(0, np.iinfo(np.uint64).max, n, dtype=np.uint64, endpoint=True)
        t = torch.from_numpy(a.view(np.int64)).cuda()
    plain = None
    for opts in ("plain", "slices=1", "slices=2", "slices=4", "slices=8", "slices=auto"):
        s = CShardedSorter(etype, 0, options=None if opts == "plain" else opts + ",loopback=1")   # plain: one rank's shortcut, copy + local sort
        for _ in range(10 if opts == "slices=auto" else 3):      # (auto tries 4, 2, 1, 8 slices twice each first)
            out, m = s.sort(t)
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            out, m = s.sort(t)
        s.check()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        if opts == "plain":
            plain = ms
        x = s.ss.exchange()
        ph = s.ss.phase_ms()
        got = out[:m].cpu().numpy().view(a.dtype)
        ok = m == n and bool(np.all(got[:-1] <= got[1:])) and int(got.sum(dtype=np.uint64)) == int(a.sum(dtype=np.uint64))
        print("2^%d %s, %-11s: %.3f ms per call (%.0f Mkeys/s, %.2fx plain)%s; slices used %d, exchange %.3f ms; phases %s" % (
            logn, etype, opts, ms, n / ms / 1e3, ms / plain, "" if ok else " WRONG RESULT", x["slices"], x["ms"],
            {k: round(v, 3) for k, v in ph.items()}), flush=True)
        s.close()
