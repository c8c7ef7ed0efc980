"""Developer fuzz (GPU box): random sizes / types / directions through the C API,
identity keys against numpy's sort, (key, value) pairs against the oracle's
restated networks and stable sort. python tools/fuzz_gpu.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cl_ops_amd as clo  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = clo.Context(0)
    q = clo.Queue(ctx)
    types = ["uchar", "char", "ushort", "short", "uint", "int", "ulong", "long", "float", "double"]
    bad = 0
    for c in range(cases):
        alg = ["abitonic", "sbitonic", "satradix"][int(rng.integers(0, 3))]
        et = types[int(rng.integers(0, len(types)))]
        dt = clo.api.CLO_TYPE_NP[et]
        top = int(os.environ.get("FUZZ_MAX_LOG2", "21"))
        logn = int(rng.integers(1, (top + 1) if alg != "sbitonic" else 17))
        n = int(rng.integers(1 << (logn - 1), (1 << logn) + 1))
        if np.issubdtype(dt, np.floating):
            a = ((rng.random(n) - 0.5) * 1e6).astype(dt)
        elif dt == np.uint64:
            a = rng.integers(0, np.iinfo(dt).max, n, dtype=np.uint64, endpoint=True)
        else:
            info = np.iinfo(dt)
            a = rng.integers(info.min, info.max, n, dtype=np.int64, endpoint=True).astype(dt)
        if rng.random() < 0.3:
            a[rng.integers(0, n, n // 2)] = a[0]          # many duplicates
        desc = alg != "satradix" and rng.random() < 0.4
        s = clo.Sorter(alg, ctx, et, compare="((a) < (b))" if desc else None)
        got = s.with_host_data(a, q)
        s.close()
        exp = np.sort(a)
        if desc:
            exp = exp[::-1]
        ok = np.array_equal(got, exp)
        if not ok:
            bad += 1
            print("MISMATCH", alg, et, n, "descending" if desc else "", flush=True)
    # (key, value) pairs: tie order
    for c in range(max(cases // 10, 4)):
        alg = ["abitonic", "sbitonic", "satradix"][c % 3]
        n = int(rng.integers(2, 1 << 16)) if alg == "satradix" else 1 << int(rng.integers(1, 17))   # (bitonic + partial keys: powers of two)
        keys = rng.integers(0, 64, n, dtype=np.uint64)
        e = (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        s = clo.Sorter(alg, ctx, "ulong", key_type="uint", get_key="(uint) ((x) >> 32)")
        got = s.with_host_data(e, q)
        s.close()
        if alg == "satradix":
            exp = O.stable_sort(e, key_size=4, key_shift=32)
        else:
            exp = O.sbitonic(e, key_size=4, key_shift=32)
        if not np.array_equal(got, exp):
            bad += 1
            print("MISMATCH pairs", alg, n, flush=True)
    print("fuzz: %d cases, %d mismatches" % (cases, bad), flush=True)
    q.close()
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
