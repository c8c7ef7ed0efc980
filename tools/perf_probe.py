"""Developer probe (not part of the product or the bench contract): times the
HIP path at the BASELINE sizes, per kernel family, through the C-ABI.
Usage on the GPU box: python tools/perf_probe.py [what ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cl_ops_amd as clo  # noqa: E402
from cl_ops_amd import _hip  # noqa: E402
from cl_ops_amd._hip import lib  # noqa: E402


def timed(fn, q, reps=5):
    t = clo.HipEventTimer(q)
    out = []
    for _ in range(reps):
        t.start()
        fn()
        t.stop()
        out.append(t.elapsed_ms())
    t.close()
    return out


def kernels(labels):
    return {l: _hip.timing_read(l) for l in labels}


def probe_radix(ctx, q, logn, etype, variant, radix=16, pairs=False):
    n = 1 << logn
    rng = np.random.default_rng(0)
    if etype == "uint":
        a = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        s = clo.Sorter("satradix", ctx, "uint", options="radix=%d" % radix)
    elif pairs:
        a = (rng.integers(0, 2**32, n, dtype=np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        s = clo.Sorter("satradix", ctx, "ulong", key_type="uint", get_key="(uint)((x)>>32)", options="radix=%d" % radix)
    else:
        a = rng.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2)
        s = clo.Sorter("satradix", ctx, "ulong", options="radix=%d" % radix)
    src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
    src.write(q, a)
    s.with_device_data(q, src, dst, n)  # warm-up + allocations
    q.finish()
    lib.clo_hip_timing_enable(1)
    lib.clo_hip_timing_reset()
    ms = timed(lambda: s.with_device_data(q, src, dst, n), q)
    k = kernels(["radix_pass", "radix_hist", "radix_offsets"])
    lib.clo_hip_timing_enable(0)
    lib.clo_hip_timing_reset()
    ms2 = timed(lambda: s.with_device_data(q, src, dst, n), q)
    # the same sort as bench.py times it: 20 calls back to back between ONE pair of events (no host synchronisation and no
    # idle gap between two sorts; `ms2` above are isolated calls, each started on an idle device after a host wait)
    t = clo.HipEventTimer(q)
    b2b = 1e9
    for _ in range(3):
        t.start()
        for _ in range(20):
            s.with_device_data(q, src, dst, n)
        t.stop()
        b2b = min(b2b, t.elapsed_ms() / 20)
    t.close()
    got = dst.read(q, a.dtype, n)
    ok = bool(np.all(got[:-1] <= got[1:])) if not pairs else bool(np.all((got[:-1] >> np.uint64(32)) <= (got[1:] >> np.uint64(32))))
    best = min(ms2)
    es = a.dtype.itemsize
    npass, tp = k["radix_pass"]
    print("radix %s%s 2^%d radix=%d variant=%d: back to back %.4f ms per sort = %.2f ps per key (%.0f Mkeys/s); isolated calls %.3f ms (min of %s) -> %.0f Mkeys/s; pass avg %.3f ms (%d launches) = %.2f TB/s moved, hist %.3f ms, offsets avg %.4f ms; sorted=%s"
          % (etype, "(pairs)" if pairs else "", logn, radix, variant, b2b, b2b * 1e9 / n, n / b2b / 1e3, best, ["%.3f" % x for x in ms2], n / best / 1e3,
             tp / max(npass, 1), npass, 2 * es * n / (max(tp, 1e-9) / max(npass, 1) * 1e-3) / 1e12, k["radix_hist"][1] / max(k["radix_hist"][0], 1), k["radix_offsets"][1] / max(k["radix_offsets"][0], 1), ok), flush=True)
    for b in (src, dst):
        b.close()
    s.close()


def probe_scan(ctx, q, logn, st="uint"):
    n = 1 << logn
    a = np.random.default_rng(0).integers(0, 128, n).astype(np.uint32)
    sc = clo.Scanner("blelloch", ctx, "uint", st)
    sdt = clo.api.CLO_TYPE_NP[st]
    src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, n * sdt.itemsize)
    src.write(q, a)
    sc.with_device_data(q, src, dst, n)
    q.finish()
    ms = timed(lambda: sc.with_device_data(q, src, dst, n), q, reps=10)
    best = min(ms)
    lib.clo_hip_timing_enable(1)
    lib.clo_hip_timing_reset()
    timed(lambda: sc.with_device_data(q, src, dst, n), q, reps=10)
    kc, kt = _hip.timing_read("scan")
    lib.clo_hip_timing_enable(0)
    print("scan uint->%s 2^%d: %.4f ms (%s) -> %.0f MValues/s, %.2f TB/s; kernel alone %.4f ms"
          % (st, logn, best, ["%.4f" % x for x in ms[:5]], n / best / 1e3, n * (4 + sdt.itemsize) / (best * 1e-3) / 1e12, kt / max(kc, 1)), flush=True)
    got = dst.read(q, sdt, n)
    exp = np.concatenate((np.zeros(1, np.uint64), np.cumsum(a.astype(np.uint64))[:-1])).astype(sdt)
    print("   correct:", bool(np.array_equal(got, exp)))
    for b in (src, dst):
        b.close()
    sc.close()


def probe_bitonic(ctx, q, logn, alg="abitonic"):
    n = 1 << logn
    a = np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    s = clo.Sorter(alg, ctx, "uint")
    src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
    src.write(q, a)
    s.with_device_data(q, src, dst, n)
    q.finish()
    lib.clo_hip_timing_enable(1)
    lib.clo_hip_timing_reset()
    timed(lambda: s.with_device_data(q, src, dst, n), q, reps=1)
    k = kernels(["bitonic_presort", "bitonic_tile", "bitonic_strided", "bitonic_strided2", "bitonic_step"])
    lib.clo_hip_timing_enable(0)
    lib.clo_hip_timing_reset()
    ms = timed(lambda: s.with_device_data(q, src, dst, n), q, reps=3)
    best = min(ms)
    got = dst.read(q, np.uint32, n)
    print("%s uint 2^%d: %.3f ms -> %.0f Mkeys/s; kernels %s sorted=%s" % (alg, logn, best, n / best / 1e3, k, bool(np.all(got[:-1] <= got[1:]))), flush=True)
    for b in (src, dst):
        b.close()
    s.close()


def main():
    what = sys.argv[1:] or ["radix", "scan", "bitonic"]
    ctx = clo.Context(0)
    q = clo.Queue(ctx)
    print("device:", ctx.device_name, flush=True)
    if "radix" in what:
        probe_radix(ctx, q, 28, "uint", 0)
        probe_radix(ctx, q, 24, "uint", 0)
        probe_radix(ctx, q, 28, "uint", 0, radix=256)
        probe_radix(ctx, q, 28, "uint", 0, radix=64)
    if "pairs" in what:
        probe_radix(ctx, q, 28, "ulong", 0, pairs=True)
        probe_radix(ctx, q, 28, "ulong", 0)
        probe_radix(ctx, q, 28, "ulong", 0, radix=256)
    if "scan" in what:
        probe_scan(ctx, q, 26, "uint")
        probe_scan(ctx, q, 26, "ulong")
        probe_scan(ctx, q, 28, "uint")
    if "bitonic" in what:
        probe_bitonic(ctx, q, 26, "abitonic")
        probe_bitonic(ctx, q, 20, "abitonic")
        probe_bitonic(ctx, q, 16, "sbitonic")
    q.close()
    ctx.close()


if __name__ == "__main__":
    main()
