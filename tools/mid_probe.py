"""Developer probe: per-kernel times of the chain-free radix passes across the mid sizes
(2^22 .. 2^28 uint32), where the per-key rate lags the headline. GPU box only.
usage: python tools/mid_probe.py [kind=uint|ulong|pairs] [lo] [hi]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("CLO_NO_WARMUP", "1")
import perf_probe as P  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "uint"
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 22
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 28
ctx = P.clo.Context(0)
q = P.clo.Queue(ctx)
print("CLO_RADIX_SWEEP=%s" % os.environ.get("CLO_RADIX_SWEEP"), flush=True)
for logn in range(lo, hi + 1):
    if kind == "uint":
        P.probe_radix(ctx, q, logn, "uint", 0)
    else:
        P.probe_radix(ctx, q, logn, "ulong", 0, pairs=(kind == "pairs"))
q.close()
ctx.close()
