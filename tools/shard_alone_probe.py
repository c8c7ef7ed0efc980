"""Developer probe (GPU box): the sharded sort's whole protocol on ONE rank that sends its keys to itself over RCCL
(CLO_SHARD_TEST_EXCHANGE_ALONE) — partition, count all-gather, sliced grouped send/recv on the transfer stream, slice
sorts in place on the exec stream — against the plain local sort. What a node adds is xGMI instead of a device copy.
usage: python tools/shard_alone_probe.py [log2n=28]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CLO_SHARD_TEST_EXCHANGE_ALONE"] = "1"
from cl_ops_amd.multigpu import CShardedSorter  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << logn
a = np.random.default_rng(0).integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
t = torch.from_numpy(a.view(np.int32)).cuda()
for opts in ("slices=1", "slices=2", None, "slices=8"):
    if opts == "slices=1":
        os.environ.pop("CLO_SHARD_TEST_EXCHANGE_ALONE")           # one rank, no exchange: copy + local sort
    else:
        os.environ["CLO_SHARD_TEST_EXCHANGE_ALONE"] = "1"
    s = CShardedSorter("uint", 0, options=opts)
    for _ in range(3):
        out, m = s.sort(t)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        out, m = s.sort(t)
    s.check()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    x = s.ss.exchange()
    ph = s.ss.phase_ms()
    print("2^%d uint32, %-9s: %.3f ms per call (%.0f Mkeys/s); slices used %d, exchange %.3f ms; phases %s" % (
        logn, opts or "slices=4", ms, n / ms / 1e3, x["slices"], x["ms"], {k: round(v, 3) for k, v in ph.items()}), flush=True)
    s.close()
