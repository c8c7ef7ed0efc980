import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo
from cl_ops_amd import _hip
from cl_ops_amd._hip import lib
et = sys.argv[1] if len(sys.argv) > 1 else "uint"
dt = np.uint32 if et == "uint" else np.uint64
n = 1 << 28
ctx = clo.Context(0); q = clo.Queue(ctx)
a = np.random.default_rng(0).integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
s = clo.Sorter("satradix", ctx, et)
src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
src.write(q, a)
for _ in range(2):
    s.with_device_data(q, src, dst, n)
q.finish()
lib.clo_hip_timing_enable(1); lib.clo_hip_timing_reset()
for _ in range(3):
    s.with_device_data(q, src, dst, n)
q.finish()
out = []
for lab in ("radix_pass", "radix_hist", "radix_offsets"):
    c, t = _hip.timing_read(lab)
    out.append("%s %.4f" % (lab, t / c))
print("XF=%s %s: %s" % (os.environ.get("CLO_RP_XF", "0"), et, ", ".join(out)))
