"""Developer probe: a few plain satradix sorts of 2^logn keys, to be run under `rocprofv3 --kernel-trace`
(tools/kernel_timeline.py prints the last call's kernels in start order). GPU box only.
usage: python tools/sort_once.py [logn=26] [uint|ulong|pairs] [calls=6]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("CLO_NO_WARMUP", "1")
import numpy as np  # noqa: E402
import perf_probe as P  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
kind = sys.argv[2] if len(sys.argv) > 2 else "uint"
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 6
n = 1 << logn
clo = P.clo
ctx = clo.Context(0)
q = clo.Queue(ctx)
rng = np.random.default_rng(0)
if kind == "uint":
    a = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    s = clo.Sorter("satradix", ctx, "uint", options="radix=16")
elif kind == "pairs":
    a = (rng.integers(0, 2**32, n, dtype=np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    s = clo.Sorter("satradix", ctx, "ulong", key_type="uint", get_key="(uint)((x)>>32)", options="radix=16")
else:
    a = rng.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2)
    s = clo.Sorter("satradix", ctx, "ulong", options="radix=16")
src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
src.write(q, a)
for _ in range(calls):
    s.with_device_data(q, src, dst, n)
    q.finish()
got = dst.read(q, a.dtype, n)
print("2^%d %s: sorted=%s" % (logn, kind, bool(np.all(got[:-1] <= got[1:])) if kind != "pairs" else "n/a"))
