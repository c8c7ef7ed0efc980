import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import cl_ops_amd as clo, oracle_lib as O
ctx = clo.Context(0); q = clo.Queue(ctx)
for logn in (21, 23):
    n = 1 << logn
    a = np.random.default_rng(logn).integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    s = clo.Sorter("abitonic", ctx, "uint", get_key="((x) >> 20)")
    got = s.with_host_data(a, q); s.close()
    t0 = time.time(); exp = O.sbitonic(a, key_shift=20); dt = time.time() - t0
    print("u32 key>>20 2^%d: bit-exact vs oracle network:" % logn, bool(np.array_equal(got, exp)), "(oracle %.1f s)" % dt, flush=True)
    keys = np.random.default_rng(1).integers(0, 1000, n, dtype=np.uint64)
    e = (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    s = clo.Sorter("abitonic", ctx, "ulong", key_type="uint", get_key="(uint) ((x) >> 32)")
    got = s.with_host_data(e, q); s.close()
    exp = O.sbitonic(e, key_size=4, key_shift=32)
    print("pairs 2^%d: bit-exact vs oracle network:" % logn, bool(np.array_equal(got, exp)), flush=True)
