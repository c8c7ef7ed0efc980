"""Developer probe: satradix in place vs out of place, device buffers, 2^28 uint."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo
ctx = clo.Context(0); q = clo.Queue(ctx)
n = 1 << 28
a = np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
s = clo.Sorter("satradix", ctx, "uint")
src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
t = clo.HipEventTimer(q)
for mode in ("out of place", "in place", "in place, fresh buffer each time"):
    ms = []
    for rep in range(5):
        if mode.endswith("each time"):
            src.close(); src = clo.Buffer(ctx, a.nbytes)
        src.write(q, a); q.finish()
        t.start()
        s.with_device_data(q, src, dst if mode == "out of place" else None, n)
        t.stop()
        ms.append(t.elapsed_ms())
    print("%-34s %s" % (mode, ["%.3f" % x for x in ms]), flush=True)
