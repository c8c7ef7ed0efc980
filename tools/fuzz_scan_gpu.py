"""Developer fuzz (GPU box): exclusive scans of random sizes and element / sum
types through the C API against numpy's modular cumulative sum (device-resident
and host-data paths). python tools/fuzz_scan_gpu.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cl_ops_amd as clo  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    top = int(os.environ.get("FUZZ_MAX_LOG2", "25"))
    ctx = clo.Context(0)
    q = clo.Queue(ctx)
    types = ["uchar", "char", "ushort", "short", "uint", "int", "ulong", "long"]
    bad = 0
    for c in range(cases):
        et = types[int(rng.integers(0, len(types)))]
        dt = clo.api.CLO_TYPE_NP[et]
        wider = [t for t in types if clo.api.CLO_TYPE_NP[t].itemsize >= dt.itemsize]
        st = wider[int(rng.integers(0, len(wider)))]
        sdt = clo.api.CLO_TYPE_NP[st]
        logn = int(rng.integers(0, top + 1))
        n = int(rng.integers(max(1, (1 << logn) // 2), (1 << logn) + 1))
        info = np.iinfo(dt)
        if dt == np.uint64:
            a = rng.integers(0, info.max, n, dtype=np.uint64, endpoint=True)
        else:
            a = rng.integers(info.min, info.max, n, dtype=np.int64, endpoint=True).astype(dt)
        # exclusive scan mod 2^64 of the sign-extended elements, truncated to the sum type
        wide = a.astype(np.int64).view(np.uint64) if dt != np.uint64 else a
        incl = np.cumsum(wide, dtype=np.uint64)
        exp = np.concatenate((np.zeros(1, np.uint64), incl[:-1])).astype(np.dtype("u%d" % sdt.itemsize)).view(sdt) if sdt.kind == "i" \
            else np.concatenate((np.zeros(1, np.uint64), incl[:-1])).astype(sdt)
        sc = clo.Scanner("blelloch", ctx, et, st)
        if rng.random() < 0.5:
            got = sc.with_host_data(a, q)
        else:
            src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, n * sdt.itemsize)
            src.write(q, a)
            sc.with_device_data(q, src, dst, n)
            got = dst.read(q, sdt, n)
            src.close(); dst.close()
        sc.close()
        if not np.array_equal(got, exp):
            bad += 1
            print("MISMATCH", et, st, n, flush=True)
    print("scan fuzz: %d cases, %d mismatches" % (cases, bad), flush=True)
    q.close(); ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
