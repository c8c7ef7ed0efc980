"""Developer helper: times the radix4 pass kernel under CLO_R4_XF ablation flags (results are NOT sorted)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo
from cl_ops_amd import _hip
from cl_ops_amd._hip import lib
n = 1 << 28
ctx = clo.Context(0); q = clo.Queue(ctx)
a = np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
s = clo.Sorter("satradix", ctx, "uint")
src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
src.write(q, a)
lib.clo_hip_radix_set_debug_buffer(None)   # picks CLO_R4_XF up
for _ in range(2):
    s.with_device_data(q, src, dst, n)
q.finish()
lib.clo_hip_timing_enable(1); lib.clo_hip_timing_reset()
for _ in range(3):
    s.with_device_data(q, src, dst, n)
q.finish()
c, t = _hip.timing_read("radix_pass")
print("XF=%s pass avg %.4f ms over %d" % (os.environ.get("CLO_R4_XF", "0"), t / c, c))
