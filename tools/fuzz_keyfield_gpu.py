"""Developer fuzz (GPU box): satradix / gselect / bitonic on random KEY FIELDS of the element — get_key = a shift and a
mask, any width 1..key bits — at sizes on every path of the radix sort (one launch, single-sweep passes, small and big
tiles), any numel, radix 16 and 256; result = numpy's stable sort by the field (satradix) or sorted-by-field + permutation
(bitonic). Round 3 wrote this after a 28-bit field on the single-sweep passes came out wrong with every other test green.
python tools/fuzz_keyfield_gpu.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cl_ops_amd as clo  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    ctx = clo.Context(0)
    q = clo.Queue(ctx)
    bad = 0
    sizes4 = [5, 4097, 16384, 16385, 40000, (1 << 17) + 3, (1 << 20) + 1, (1 << 22) + 77, (1 << 23) + 8193, (1 << 24) + 5, (1 << 26) + 9]
    sizes8 = [3, 4097, 8193, 30000, (1 << 16) + 3, (1 << 19) + 1, (1 << 21) + 77, (1 << 22) + 4097, (1 << 23) + 5, (1 << 25) + 9]
    for c in range(cases):
        es = int(rng.choice([4, 8]))
        et, dt = ("uint", np.uint32) if es == 4 else ("ulong", np.uint64)
        bits = 8 * es
        n = int(rng.choice(sizes4 if es == 4 else sizes8)) + int(rng.integers(0, 5))
        if rng.random() < 0.5:
            n = min(n, (1 << 21) + int(rng.integers(0, 1000)))        # most cases small enough to keep the run short
        width = int(rng.integers(1, bits + 1))
        shift = int(rng.integers(0, bits - width + 1))
        mask = (1 << width) - 1
        a = rng.integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
        if rng.random() < 0.3:
            a &= dt((((1 << int(rng.integers(1, bits))) - 1) << int(rng.integers(0, 8))) & ((1 << bits) - 1))     # constant digits here and there
        key = (a >> dt(shift)) & dt(mask)
        kt = "uint" if width <= 32 else "ulong"
        get_key = "(%s) (((x) >> %d) & 0x%x%s)" % (kt, shift, mask, "ul" if es == 8 else "u") if rng.random() < 0.7 or shift + width < bits \
            else "(%s) ((x) >> %d)" % (kt, shift)
        alg = str(rng.choice(["satradix", "satradix", "satradix", "abitonic", "sbitonic"]))
        if alg != "satradix":
            n = min(n, 200000)
            a, key = a[:n], key[:n]
        opts = "radix=%d" % int(rng.choice([16, 256, 4])) if alg == "satradix" else None
        try:
            s = clo.Sorter(alg, ctx, et, key_type=kt, get_key=get_key, options=opts)
            got = s.with_host_data(a, q)
            s.close()
        except clo.CloError as e:
            bad += 1
            print("ERROR", alg, et, n, get_key, opts, e.message, flush=True)
            continue
        if alg == "satradix":
            ok = np.array_equal(got, a[np.argsort(key, kind="stable")])
        else:
            gk = (got >> dt(shift)) & dt(mask)
            ok = bool(np.all(gk[:-1] <= gk[1:])) and np.array_equal(np.sort(got), np.sort(a))
        if not ok:
            bad += 1
            print("MISMATCH", alg, et, n, get_key, opts, flush=True)
        if c % 25 == 24:
            print("... %d cases, %d bad" % (c + 1, bad), flush=True)
    print("key-field fuzz: %d cases, %d mismatches" % (cases, bad), flush=True)
    q.close()
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
