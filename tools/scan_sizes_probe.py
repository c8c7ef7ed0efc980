"""Developer probe (GPU box): device-resident scan times over sizes, uint -> uint and uint -> ulong.
usage: python tools/scan_sizes_probe.py [lo=18] [hi=27]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("CLO_NO_WARMUP", "1")
import perf_probe as P  # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 18
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 27
ctx = P.clo.Context(0)
q = P.clo.Queue(ctx)
for st in ("uint", "ulong"):
    for logn in range(lo, hi + 1):
        P.probe_scan(ctx, q, logn, st)
